"""MI355X-native engine for the training hot path of
YukiYasuda2718/3d-sr-micrometeorology (3-D voxel super-resolution U-Net).

Importing this package loads ``libsr3d.so`` (hand-written HIP for gfx950); there
is no CPU or ATen fallback -- a missing library is an ImportError-time failure.

The directory name is not a Python identifier; import it with
``importlib.import_module("3d-sr-micrometeorology_amd")`` or through the
``sr3d_amd`` alias module at the repository root.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from . import ops  # noqa: F401
from .model.unet import UNetSR  # noqa: F401
from .src.loss_maker import make_loss  # noqa: F401
from .src.model_maker import make_model  # noqa: F401
from .src.optim import FlatAdam  # noqa: F401
from .src.ddp import GradAllReducer  # noqa: F401
from .src.graph import GraphedTrainStep  # noqa: F401

__version__ = "0.1.0"
