from .averaging import Averaging, GradReducer, ModelEMA, get_param_groups  # noqa: F401
