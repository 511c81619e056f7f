"""Reference-compatible `from config import *` surface (reference config/__init__.py)."""
from .ppyolo_2x import PPYOLO_2x_Config
from .ppyolo_r18vd import PPYOLO_r18vd_Config
from .get_model import select_backbone, select_head, select_loss, select_optimizer
