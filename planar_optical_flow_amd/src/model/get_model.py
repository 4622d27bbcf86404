"""``get_model(cfg)`` of the reference (src/model/get_model.py:1-6)."""


def get_model(cfg):
    if cfg["type"] == "box_reg":
        from .box_regression import BoundingBoxRegressor
        return BoundingBoxRegressor(cfg)
    raise NotImplementedError(cfg["type"])
