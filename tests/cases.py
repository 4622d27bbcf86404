"""Shared parameter sets for the oracle and HIP parity tests."""
CUTOUT_CASES = {
    "config_test": (0.5, 450, dict(fixed=False, centered=True, window_width=1.0, window_depth=0.5,
                                   num_cutout_pts=56, padding_val=29.99, area_mode=True)),
    "dr_spaam": (0.5, 450, dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5,
                                num_cutout_pts=56, padding_val=29.99, area_mode=True)),
    "defaults": (0.5, 450, dict(centered=False)),
    "stride2": (0.5, 450, dict(stride=2, fixed=True, window_width=1.3, window_depth=0.7,
                               num_cutout_pts=32, padding_val=29.99, area_mode=True)),
    "dense3600": (0.1, 3600, dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5,
                                  num_cutout_pts=56, padding_val=29.99, area_mode=True)),
    "near": (0.5, 450, dict(fixed=True, centered=True, window_width=1.0, window_depth=0.5,
                            num_cutout_pts=56, padding_val=29.99, area_mode=True)),
}
