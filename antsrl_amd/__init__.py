"""antsrl_amd — MI355X-native AntsRL environment step loop (hand-written HIP behind the
reference's RLApi surface).  See DESIGN.md."""
from . import config  # noqa: F401
from .config import AntsCfg, make_cfg  # noqa: F401

__all__ = ["config", "AntsCfg", "make_cfg"]
