"""Dense-flow stage handle (sind_flow_* in include/sind_hip.h): batched DeepFlow + VariationalRefinement on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr


class FlowStage:
    def __init__(self, fw: int = 384, fh: int = 288, max_batch: int = 8, device: int = 0):
        self.fw, self.fh, self.max_batch = fw, fh, max_batch
        h = C.c_void_p()
        check(lib().sind_flow_create(fw, fh, max_batch, device, C.byref(h)), "sind_flow_create")
        self._h = h

    def set_max_levels(self, n: int):
        """finest n levels of the 0.95 pyramid only (BASELINE.json config 5, "3-level flow pyramid"); 0 = the full pyramid of OpenCV's DeepFlow"""
        check(lib().sind_flow_set_max_levels(self._h, int(n)), "sind_flow_set_max_levels")

    def set_sor_variant(self, mode: int = 4, fuse: int = 5, tile_w: int = 64, tile_h: int | None = None):
        """solver variant of this handle (all return the same bits).  Fused register-resident SOR with 1x8 strips: mode 4 = divisions through a reciprocal formed on the fly
        (default), 5 = the streaming kernel wherever it fits, 6 = the one-wave pipeline on every level beyond one workgroup, 0 = one launch per colour (cross-check); lab builds: 1 = IEEE division, 3 = reciprocal planes held in registers
        (three waves per SIMD; 256/384/768-thread tiles), 2 = 1x4 strips and reciprocal division.  fuse = iterations per launch on the tiled levels, 0 = a per-level plan
        (lab builds).  tile_h defaults to 48 (mode 3) / 64."""
        if tile_h is None:
            tile_h = 48 if mode == 3 else 64
        check(lib().sind_flow_set_sor_tiled(self._h, mode, fuse, tile_w, tile_h), "sind_flow_set_sor_tiled")

    def set_solver_workgroups(self, cap: int):
        check(lib().sind_flow_set_solver_workgroups(self._h, int(cap)), "sind_flow_set_solver_workgroups")

    def set_wave_solver(self, on: bool = True, target_items: int = 0, bands: int = 0):
        """one-wave row pipelines (k_sor_wave) where the streaming kernel would run (default on); target_items: waves per launch the row bands are cut for (0 = default),
        bands > 0: exactly that many row bands (tests); same bits"""
        check(lib().sind_flow_set_wave_solver(self._h, 1 if on else 0, int(target_items), int(bands)), "sind_flow_set_wave_solver")

    def set_coef_kernel(self, variant: int):
        check(lib().sind_flow_set_coef_kernel(self._h, int(variant)), "sind_flow_set_coef_kernel")

    def set_coarse_chain(self, on: bool):
        """the one-workgroup pyramid levels in one launch (default) or through the per-stage kernels (cross-check); same bits"""
        check(lib().sind_flow_set_coarse_chain(self._h, 1 if on else 0), "sind_flow_set_coarse_chain")

    def set_level_up(self, on: bool):
        """level transition (W += dW, up-sampling, next level's warp) in one launch (default) or three (cross-check); same bits"""
        check(lib().sind_flow_set_level_up(self._h, 1 if on else 0), "sind_flow_set_level_up")

    def set_latency_tiles(self, on: bool):
        """tiled levels of few images through the 1024-thread tiles with deep halos (default) or the throughput kernels (cross-check); same bits"""
        check(lib().sind_flow_set_latency_tiles(self._h, 1 if on else 0), "sind_flow_set_latency_tiles")

    def close(self):
        if getattr(self, "_h", None):
            lib().sind_flow_destroy(self._h); self._h = None

    __del__ = close

    def levels(self):
        ws = np.zeros(256, np.int32); hs = np.zeros(256, np.int32)
        n = check(lib().sind_flow_levels(self._h, ptr(ws), ptr(hs), 256)); return list(zip(ws[:n].tolist(), hs[:n].tolist()))

    def deepflow(self, i0: np.ndarray, i1: np.ndarray):
        """i0, i1: u8 [B, fh, fw] -> (u, v) f32 [B, fh, fw] (raw DeepFlow output, not negated)."""
        i0 = np.ascontiguousarray(i0, np.uint8); i1 = np.ascontiguousarray(i1, np.uint8)
        B = i0.shape[0]; u = np.empty(i0.shape, np.float32); v = np.empty(i0.shape, np.float32)
        check(lib().sind_flow_deepflow(self._h, ptr(i0), ptr(i1), B, ptr(u), ptr(v)), "sind_flow_deepflow"); return u, v

    def refine(self, i0, i1, u, v):
        i0 = np.ascontiguousarray(i0, np.uint8); i1 = np.ascontiguousarray(i1, np.uint8)
        u = np.array(u, np.float32, copy=True, order="C"); v = np.array(v, np.float32, copy=True, order="C")
        check(lib().sind_flow_refine(self._h, ptr(i0), ptr(i1), i0.shape[0], ptr(u), ptr(v)), "sind_flow_refine"); return u, v

    def varref_f32(self, i0, i1, u, v, fp_iters=5, sor_iters=5, alpha=20.0, delta=5.0, gamma=10.0, omega=1.6):
        i0 = np.ascontiguousarray(i0, np.float32); i1 = np.ascontiguousarray(i1, np.float32)
        u = np.array(u, np.float32, copy=True, order="C"); v = np.array(v, np.float32, copy=True, order="C")
        B, h, w = i0.shape
        check(lib().sind_flow_varref_f32(self._h, ptr(i0), ptr(i1), w, h, B, ptr(u), ptr(v), fp_iters, sor_iters, C.c_float(alpha),
                                         C.c_float(delta), C.c_float(gamma), C.c_float(omega)), "sind_flow_varref_f32"); return u, v

    # device-pointer, asynchronous variants (bench / pipeline)
    def deepflow_dev(self, i0_ptr: int, i1_ptr: int, B: int, u_ptr: int, v_ptr: int):
        check(lib().sind_flow_deepflow_dev(self._h, ptr(i0_ptr), ptr(i1_ptr), B, ptr(u_ptr), ptr(v_ptr)), "sind_flow_deepflow_dev")

    def refine_dev(self, i0_ptr, i1_ptr, B, u_ptr, v_ptr):
        check(lib().sind_flow_refine_dev(self._h, ptr(i0_ptr), ptr(i1_ptr), B, ptr(u_ptr), ptr(v_ptr)), "sind_flow_refine_dev")

    def sync(self):
        check(lib().sind_flow_sync(self._h), "sind_flow_sync")

    def timer_begin(self):
        check(lib().sind_flow_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float(); check(lib().sind_flow_timer_end(self._h, C.byref(ms))); return ms.value
