"""MI355X-native NDT scan matcher: thin host-side mirror of the C ABI in include/ndt_hip.h.

    matcher   NdtMatcher2D / NdtBatch2D / NdtMulti2D / NdtMatcher3D / NdtPyramid2D (ctypes over
              gtsam_ndt_amd/lib/libndt_hip.so; there is no CPU fallback - loading fails loudly
              when the HIP library has not been built: python -c 'import __graft_entry__ as g; g.build()')
    dist      one process per GPU: pair sharding and the RCCL gather of the result rows
    synth     deterministic synthetic scans (BASELINE configs 1-4, planar-lidar ray caster)
    synth3d   config 5 (64-beam 3D lidar)
    build     hipcc / gcc recipes

The CPU oracle lives in the top-level `oracle` package and is test infrastructure only.
"""
__all__ = ["matcher", "dist", "synth", "synth3d", "build"]
