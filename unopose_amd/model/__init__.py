from .config import Cfg, default_model_cfg  # noqa: F401
from .unopose import UNOPose  # noqa: F401
