"""Device-resident problems for the HIP VB engine, with torch used only as the allocator /
stream / collective plumbing (the engine itself takes raw device pointers).

    prob = DeviceProblem(holder, data, device)   # uploads once
    prob.run()                                   # one asynchronous pass of the hot path
    res = prob.results()                         # numpy copies
"""
import copy
import ctypes as C

import numpy as np
import torch

from . import hiplib, vbabi


class DeviceProblem:
    def __init__(self, holder, data, device="cuda:0", want_f=None):
        self.holder = holder
        cfg = holder.cfg
        self.device = torch.device(device)
        if isinstance(data, torch.Tensor):  # a series that was generated on the device (full-size tests)
            assert data.is_cuda and data.is_contiguous() and data.dtype in (torch.float32, torch.float64)
            assert tuple(data.shape) == (cfg.n_times, cfg.n_voxels)
            self.data = data
            is_f64 = data.dtype == torch.float64
        else:
            data = np.ascontiguousarray(data)
            if data.dtype != np.float64:
                data = np.ascontiguousarray(data, dtype=np.float32)
            assert data.shape == (cfg.n_times, cfg.n_voxels)
            self.data = torch.from_numpy(data).to(self.device)
            is_f64 = data.dtype == np.float64
        # device copy of the config: same scalars, pointer members replaced by device tensors
        self.cfg = vbabi.FvbConfig.from_buffer_copy(cfg)
        self.cfg.data_f64 = 1 if is_f64 else 0
        self._keep = {}
        for name, arr in holder.keep.items():
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
            self._keep[name] = t
            if name == "design":
                self.cfg.design = t.data_ptr()
            elif name == "phi_index":
                self.cfg.phi_index = t.data_ptr()
            elif name == "init_mvn":
                self.cfg.init_mvn = t.data_ptr()
            elif name.startswith("image_"):
                self.cfg.image_prior[int(name.split("_")[1])] = t.data_ptr()
        V = cfg.n_voxels
        self.mvn = torch.empty((holder.n_mvn_rows, V), dtype=torch.float64, device=self.device)
        self.free_energy = torch.full((V,), float("nan"), dtype=torch.float64, device=self.device)  # (voxels that fail before any F keep NaN)
        self.status = torch.empty(V, dtype=torch.int32, device=self.device)
        self.iterations = torch.empty(V, dtype=torch.int32, device=self.device)
        self.out = vbabi.FvbOutputs()
        self.out.mvn = self.mvn.data_ptr()
        self.out.free_energy = self.free_energy.data_ptr()
        self.out.status = self.status.data_ptr()
        self.out.iterations = self.iterations.data_ptr()
        self.f_history = None
        if cfg.f_history_rows > 0:
            self.f_history = torch.full((cfg.f_history_rows, V), float("nan"), dtype=torch.float64, device=self.device)
            self.f_history_len = torch.zeros(V, dtype=torch.int32, device=self.device)
            self.out.f_history = self.f_history.data_ptr()
            self.out.f_history_len = self.f_history_len.data_ptr()
        self.n_unmasked = hiplib.n_unmasked(holder)
        self.kernel = hiplib.kernel_name(holder)

    def run(self, stream=None):
        """Enqueue one pass of the voxelwise VB loop on `stream` (default: torch's current)."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        rc = hiplib.lib().fabber_vb_run_device_ex(C.byref(self.cfg), self.data.data_ptr(), C.byref(self.out),
                                                  C.c_void_p(stream.cuda_stream), self.n_unmasked)
        if rc != 0:
            raise hiplib.HipEngineError("fabber_vb_run_device_ex: %d %s" % (rc, hiplib.lib().fabber_vb_last_error().decode()))

    def run_spatial(self, spatial, stream=None):
        """One complete spatial VB run (all max_iterations sweeps) on `stream`; returns after the
        stream has drained (the driver frees its work buffers on return). spatial: SpatialHolder."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        L = hiplib.lib()
        L.fabber_vb_run_spatial_device.restype = C.c_int32
        L.fabber_vb_run_spatial_device.argtypes = [C.POINTER(vbabi.FvbConfig), C.POINTER(vbabi.FvbSpatial), C.c_void_p,
                                                   C.POINTER(vbabi.FvbOutputs), C.c_void_p, C.c_void_p]
        rc = L.fabber_vb_run_spatial_device(C.byref(self.cfg), C.byref(spatial.sp), self.data.data_ptr(), C.byref(self.out),
                                            C.c_void_p(stream.cuda_stream), None)
        if rc != 0:
            raise hiplib.HipEngineError("fabber_vb_run_spatial_device: %d %s" % (rc, L.fabber_vb_last_error().decode()))

    def results(self):
        torch.cuda.synchronize(self.device)
        st = self.status.cpu().numpy()
        res = dict(mvn=self.mvn.cpu().numpy(), free_energy=self.free_energy.cpu().numpy(),
                   status=st & 0xFF, setup_failed=(st & 0x100) != 0, iterations=self.iterations.cpu().numpy())
        if self.f_history is not None:
            res["f_history"] = self.f_history.cpu().numpy()
            res["f_history_len"] = self.f_history_len.cpu().numpy()
        return res
