"""unet_amd: MI355X-native kernels behind the reference's UNet / Swin-UNet / loss / tiled-predict API."""
from .unet import OptimizedUNet  # noqa: F401
from .losses import MultiLossFunction, TIDWeightedLoss, GradientLoss, SpectralLoss  # noqa: F401
from .swin import SwinUNet  # noqa: F401
from .predictor import LargeAreaPredictor, LargeAreaPredictionConfig  # noqa: F401
from .optim import FusedAdamW, clip_grad_norm_  # noqa: F401
from .data import EnhancedDataPreparation  # noqa: F401
from .metrics import StreamingMetrics  # noqa: F401
from . import graph, prep  # noqa: F401
