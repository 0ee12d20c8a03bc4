from .cerberus import CerberusDet, Controller  # noqa: F401
from .common import C2f, Concat, Conv, SPPF, Bottleneck, Upsample  # noqa: F401
from .yolo import DFL, Detect, Model  # noqa: F401
