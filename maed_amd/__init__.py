"""maed_amd: MI355X-native (gfx950) implementation of the MAED forward/backward hot path.

Host side = PyTorch-ROCm nn.Modules with the reference's `lib.models` API; compute = hand-written HIP
kernels in libmaed_hip.so (C-ABI: include/maed_hip.h).  See DESIGN.md / INTEGRATION.md.
"""
from .maed import MAED  # noqa: F401
from .ktd import KTD  # noqa: F401
from .vision_transformer import VisionTransformer, Block, Attention, Mlp, vit_custom_resnet50_224_in21k  # noqa: F401
from .resnetv2 import ResNetV2  # noqa: F401
from .smpl import SMPL  # noqa: F401
from .iterative import Regressor  # noqa: F401
from .evaluate import Evaluator  # noqa: F401
from .ops import set_float32_matmul_precision, get_float32_matmul_precision, set_float32_backward_precision, get_float32_backward_precision  # noqa: F401
from ._lib import device_faults  # noqa: F401,E402
