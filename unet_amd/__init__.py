"""unet_amd: MI355X-native kernels behind the reference's UNet / Swin-UNet / loss / tiled-predict API."""
from .unet import OptimizedUNet  # noqa: F401
