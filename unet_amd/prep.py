"""Long-wavelength background on the GPU: the Gaussian low-pass of the data-preparation stage.

Drop-ins for three methods of the reference's ``EnhancedBathymetryProcessor`` (data_preparation.py:246):

    calculate_sigma_pixels(cutoff_wavelength_km, data_array)      :365-390   (host arithmetic)
    apply_gaussian_filter(data, sigma_lat, sigma_lon)             :392-414   (NaN -> 0, filter, NaN back)
    generate_longwave_background(bathy_true)                      :416-434

``scipy.ndimage.gaussian_filter(x, sigma=[s_lat, s_lon], mode='reflect')`` becomes two passes of a
float64 row filter around two transposes, all on the device; weights and radius follow scipy
(radius = int(4 sigma + 0.5), normalised exp(-x^2 / 2 sigma^2)).  Results agree with scipy to
rounding (the summation order differs); NaN positions are identical.
"""
from typing import Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream

TRUNCATE = 4.0     # scipy.ndimage.gaussian_filter default


def gaussian_kernel1d(sigma: float) -> np.ndarray:
    """scipy.ndimage's order-0 kernel: exp(-0.5 x^2 / sigma^2) on [-r, r], r = int(4 sigma + 0.5), sum 1."""
    sigma = float(sigma)
    radius = int(TRUNCATE * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def calculate_sigma_pixels(cutoff_wavelength_km: float, lat_coords, lon_coords) -> Tuple[float, float]:
    """data_preparation.py:365-390 on coordinate arrays (``data_array['lat'].values`` / ``['lon'].values``)."""
    lat_coords, lon_coords = np.asarray(lat_coords), np.asarray(lon_coords)
    d_lat = np.abs(float(np.mean(np.diff(lat_coords))))
    d_lon = np.abs(float(np.mean(np.diff(lon_coords))))
    mean_lat_rad = np.radians(np.mean(lat_coords))
    km_per_deg_lat = 111.0
    km_per_deg_lon = 111.0 * np.cos(mean_lat_rad)
    cutoff_lat_pixels = cutoff_wavelength_km / km_per_deg_lat / d_lat
    cutoff_lon_pixels = cutoff_wavelength_km / km_per_deg_lon / d_lon
    return cutoff_lat_pixels / 2.355, cutoff_lon_pixels / 2.355


def _padded_weights(sigma: float, device) -> Tuple[torch.Tensor, int]:
    w = gaussian_kernel1d(sigma)[::-1]        # correlation weights (symmetric: the flip is scipy's, a no-op)
    radius = (len(w) - 1) // 2
    rmax = _lib.load().unetk_gauss_rows_max_radius()
    if radius > rmax:
        raise ValueError(f"sigma {sigma:.2f} pixels needs radius {radius}; the filter kernel holds {rmax}")
    n = -(-len(w) // 16) * 16
    wp = np.zeros(n, np.float64)
    wp[:len(w)] = w
    return torch.from_numpy(wp).to(device), radius


def gaussian_filter_device(x: torch.Tensor, sigma_lat: float, sigma_lon: float, nan_aware: bool = True) -> torch.Tensor:
    """x: (lat, lon) float64 CUDA tensor -> filtered tensor of the same shape.  nan_aware: NaNs count as
    0 in the sums and come back as NaN in the result (the reference's apply_gaussian_filter)."""
    if not x.is_cuda:
        raise RuntimeError("gaussian_filter_device runs on the GPU only (no CPU fallback)")
    if x.dim() != 2 or x.dtype != torch.float64:
        raise ValueError("expected a 2-D float64 tensor")
    x = x.contiguous()
    H, W = x.shape
    dev = x.device
    w_lat, r_lat = _padded_weights(sigma_lat, dev)
    w_lon, r_lon = _padded_weights(sigma_lon, dev)
    # scipy filters axis 0 (lat) first, then axis 1 (lon)
    t1 = torch.empty((W, H), dtype=torch.float64, device=dev)
    call("unetk_transpose_f64", ptr(x), ptr(t1), H, W, stream())
    f1 = torch.empty_like(t1)
    call("unetk_gauss_rows_f64", ptr(t1), ptr(f1), None, W, H, ptr(w_lat), r_lat, 1 if nan_aware else 0, stream())
    t2 = torch.empty((H, W), dtype=torch.float64, device=dev)
    call("unetk_transpose_f64", ptr(f1), ptr(t2), W, H, stream())
    out = torch.empty_like(t2)
    call("unetk_gauss_rows_f64", ptr(t2), ptr(out), ptr(x) if nan_aware else None, H, W, ptr(w_lon), r_lon, 0, stream())
    return out


def apply_gaussian_filter(data, sigma_lat: float, sigma_lon: float, device=None):
    """data_preparation.py:392-414.  ``data``: a 2-D array (numpy, or anything with ``.values``) or a CUDA
    tensor; returns float64 of the same kind (numpy in -> numpy out)."""
    if isinstance(data, torch.Tensor):
        return gaussian_filter_device(data.double(), sigma_lat, sigma_lon)
    values = np.asarray(getattr(data, "values", data)).astype(float)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("apply_gaussian_filter runs on the GPU only (no CPU fallback)")
    out = gaussian_filter_device(torch.from_numpy(values).to(dev), sigma_lat, sigma_lon)
    return out.cpu().numpy()


def generate_longwave_background(bathy_true, lat_coords, lon_coords, long_cutoff_wavelength: float = 135.0,
                                 device=None):
    """data_preparation.py:416-434: the >= cutoff-km component; returns (background, sigma_lat, sigma_lon)."""
    sigma_lat, sigma_lon = calculate_sigma_pixels(long_cutoff_wavelength, lat_coords, lon_coords)
    return apply_gaussian_filter(bathy_true, sigma_lat, sigma_lon, device), float(sigma_lat), float(sigma_lon)
