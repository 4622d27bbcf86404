// N1, host side: numeric CSV -> float64 matrix.
//
// The reference parses every DROW sequence with np.genfromtxt(delimiter=",")
// (src/utils/dataset_dr_spaam.py:473-478 `<seq>.csv`, :504-509 `<seq>.odom2`,
// :497-502 `<seq>.difodom`, bin/data_prepare.py:70-72 `<seq>.flow`), which
// converts each field with Python's float(): a correctly rounded decimal ->
// binary64 conversion.  glibc strtod is correctly rounded as well, so this
// parser returns bit-identical doubles; the narrowing casts (uint32 sequence
// numbers, float32 times / ranges) stay with the caller, as in the reference.
//
// No device code: the file is scanned once for line starts, then the lines are
// converted by a few host threads straight into the caller's buffer.
#include "pof_common.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char *path)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = static_cast<const char *>(m);
        return true;
    }
    ~Mapped()
    {
        if (p) munmap(const_cast<char *>(p), n);
        if (fd >= 0) ::close(fd);
    }
};

inline bool blank_line(const char *b, const char *e)
{
    for (; b < e; ++b)
        if (*b != ' ' && *b != '\t' && *b != '\r') return false;
    return true;
}

// start offsets of the non-blank, non-comment lines (genfromtxt skips both)
void line_starts(const Mapped &f, std::vector<size_t> &starts, std::vector<size_t> &ends)
{
    size_t b = 0;
    while (b < f.n) {
        const char *nl = static_cast<const char *>(memchr(f.p + b, '\n', f.n - b));
        const size_t e = nl ? (size_t)(nl - f.p) : f.n;
        if (!blank_line(f.p + b, f.p + e) && f.p[b] != '#') {
            starts.push_back(b);
            ends.push_back(e);
        }
        b = e + 1;
    }
}

int count_fields(const char *b, const char *e)
{
    int c = 1;
    for (; b < e; ++b) c += (*b == ',');
    return c;
}

// one field [b, e): genfromtxt strips blanks; an empty or unparsable field is nan
double parse_field(const char *b, const char *e)
{
    while (b < e && (*b == ' ' || *b == '\t')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
    const size_t len = (size_t)(e - b);
    if (len == 0) return NAN;
    char buf[64];
    if (len < sizeof buf) {
        memcpy(buf, b, len);
        buf[len] = 0;
        char *end = nullptr;
        const double v = strtod(buf, &end);
        return (end == buf + len) ? v : NAN;
    }
    std::vector<char> big(b, e);
    big.push_back(0);
    char *end = nullptr;
    const double v = strtod(big.data(), &end);
    return (end == big.data() + len) ? v : NAN;
}

}  // namespace

extern "C" int pof_csv_shape(const char *path, long long *rows, int *cols)
{
    if (!path || !rows || !cols) return POF_E_BADARG;
    Mapped f;
    if (!f.open(path)) return POF_E_BADARG;
    std::vector<size_t> st, en;
    line_starts(f, st, en);
    *rows = (long long)st.size();
    *cols = st.empty() ? 0 : count_fields(f.p + st[0], f.p + en[0]);
    return POF_OK;
}

extern "C" int pof_csv_read_f64(const char *path, long long rows, int cols, double *out, int threads)
{
    if (!path || !out || rows < 0 || cols < 1) return POF_E_BADARG;
    Mapped f;
    if (!f.open(path)) return POF_E_BADARG;
    std::vector<size_t> st, en;
    line_starts(f, st, en);
    if ((long long)st.size() != rows) return POF_E_SHAPE;
    // every row must have the announced number of columns (genfromtxt raises otherwise)
    std::atomic<int> bad{0};
    auto work = [&](long long r0, long long r1) {
        for (long long r = r0; r < r1; ++r) {
            const char *b = f.p + st[r], *e = f.p + en[r];
            int c = 0;
            while (true) {
                const char *comma = static_cast<const char *>(memchr(b, ',', (size_t)(e - b)));
                const char *fe = comma ? comma : e;
                if (c < cols) out[r * cols + c] = parse_field(b, fe);
                ++c;
                if (!comma) break;
                b = comma + 1;
            }
            if (c != cols) bad.store(1, std::memory_order_relaxed);
        }
    };
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if (rows < 64) nt = 1;
    if (nt == 1) {
        work(0, rows);
    } else {
        std::vector<std::thread> pool;
        const long long per = (rows + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            const long long r0 = t * per, r1 = r0 + per < rows ? r0 + per : rows;
            if (r0 < r1) pool.emplace_back(work, r0, r1);
        }
        for (auto &th : pool) th.join();
    }
    return bad.load() ? POF_E_SHAPE : POF_OK;
}


// ---------------------------------------------------------------------------------------
// LZF block decoder for `DATA binary_compressed` .pcd payloads (the reference reaches liblzf
// through python-lzf in src/data_handle/_pypcd.py:249-264).  Control byte c < 32: c + 1 literal
// bytes follow; otherwise a back reference of (c >> 5) + 2 bytes (a 7 in the top bits adds the
// next byte to the length) at distance ((c & 31) << 8 | next byte) + 1.  Every read and write is
// bounds checked; a reference may overlap its own output (run-length use), so it is copied bytewise.
// ---------------------------------------------------------------------------------------
extern "C" long long pof_lzf_decompress(const void *in_, long long in_len, void *out_, long long out_cap)
{
    if ((!in_ && in_len) || (!out_ && out_cap) || in_len < 0 || out_cap < 0) return -1;
    const unsigned char *ip = static_cast<const unsigned char *>(in_), *ie = ip + in_len;
    unsigned char *out = static_cast<unsigned char *>(out_);
    long long op = 0;
    while (ip < ie) {
        unsigned c = *ip++;
        if (c < 32) {
            long long run = c + 1;
            if (ie - ip < run || out_cap - op < run) return -1;
            std::memcpy(out + op, ip, run);
            ip += run;
            op += run;
        } else {
            long long len = c >> 5;
            if (len == 7) {
                if (ip >= ie) return -1;
                len += *ip++;
            }
            if (ip >= ie) return -1;
            long long dist = ((long long)(c & 31) << 8 | *ip++) + 1;
            len += 2;
            if (dist > op || out_cap - op < len) return -1;
            const unsigned char *ref = out + op - dist;
            for (long long k = 0; k < len; ++k) out[op + k] = ref[k];
            op += len;
        }
    }
    return op;
}
