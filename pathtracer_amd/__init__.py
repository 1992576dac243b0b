"""MI355X-native ray-tracing core for the hot path of JovianGraphics/PathTracer.

Everything computes in pathtracer_amd/libptcore.so (hand-written HIP for gfx950 behind the C ABI
of include/ptcore.h); this package is the thin host-side mirror of the reference's scene
preparation and dispatch sites.  There is no CPU fallback: importing works anywhere, but creating
a `PathTracer` context needs the built library and a gfx950 device.
"""
from .core import (AUX_DT, CONST_DT, HIT_DT, LIGHT_DT, NODE2_DT, NODE_DT, RAY_DT, SLOT_DT, TRI_DT, PathTracer, PtError,
                   build_bvh, default_lights, fill_constants, flatten_bvh, flatten_bvh_compact, flatten_bvh_treelets, load_blue_noise, load_ply, lib_path,
                   read_frame_dump, write_frame_dump, write_ppm)
from . import scenes

__all__ = ["PathTracer", "PtError", "build_bvh", "flatten_bvh", "flatten_bvh_compact", "flatten_bvh_treelets", "load_ply", "fill_constants", "load_blue_noise",
           "default_lights", "lib_path", "scenes", "read_frame_dump", "write_frame_dump", "write_ppm", "NODE_DT", "NODE2_DT", "TRI_DT", "RAY_DT", "HIT_DT", "LIGHT_DT",
           "AUX_DT", "SLOT_DT", "CONST_DT"]
