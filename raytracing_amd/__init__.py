"""MI355X-native render loop for the RTOW sphere path tracer.

Host-side Python mirror of the C ABI in include/rt_amd.h (librt_amd.so, hand-written HIP for
gfx950).  The library is the product; this package only loads it, marshals plain arrays and
splits the image into bands across ranks.  There is no CPU fallback: if the HIP library is
missing or no GPU is visible, calls raise.
"""
from ._lib import (scene_grid, render_multi, RtError, Camera, CameraPod, Params, Renderer, device_count, format_ppm, load_library,
                   write_ppm, write_ppm_binary, MODE_MEGAKERNEL, MODE_WAVEFRONT, MODE_PIXEL, MODE_BVH, LAMBERTIAN, METAL, DIELECTRIC)
from .scenes import final_scene_table, three_sphere_table, final_camera, three_camera, make_camera

__all__ = ["scene_grid", "render_multi", "RtError", "Camera", "CameraPod", "Params", "Renderer", "device_count", "format_ppm", "load_library",
           "write_ppm", "write_ppm_binary", "MODE_MEGAKERNEL", "MODE_WAVEFRONT", "MODE_PIXEL", "MODE_BVH", "LAMBERTIAN", "METAL", "DIELECTRIC",
           "final_scene_table", "three_sphere_table", "final_camera", "three_camera", "make_camera"]
