"""
dist.py — data-parallel plumbing for the SVI step: one process per GPU, the global
minibatch sharded by contiguous slices, ONE all-reduce(SUM) per step over the flat
gradient buffer with the ELBO scalars in its last 4 slots (RCCL over xGMI when the
backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code; sharding is exact because its
loss is a SUM over the plate (models/ivae.py:177,215): the global gradient is the
sum of the shard gradients, so no averaging is applied.
"""
import ctypes as C
import os
import sys as _sys
from typing import Optional, Tuple

import torch
import torch.distributed as td


def world(group=None) -> Tuple[int, int]:
    """(rank, world_size) of the process group, (0, 1) when torch.distributed is not initialised."""
    if td.is_available() and td.is_initialized():
        return td.get_rank(group), td.get_world_size(group)
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a global batch of n samples owned by `rank`.
    The first n % world_size ranks get one extra sample; ranks may get an empty slice."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allreduce_sum_(flat: torch.Tensor, group=None) -> None:
    """In-place SUM all-reduce of the flat gradient(+scalars) buffer; no-op for world size 1."""
    if td.is_available() and td.is_initialized() and td.get_world_size(group) > 1:
        td.all_reduce(flat, op=td.ReduceOp.SUM, group=group)


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> None:
    """Makes every replica start from rank `src`'s parameters."""
    if td.is_available() and td.is_initialized() and td.get_world_size(group) > 1:
        td.broadcast(flat, src=src, group=group)


def sync_replicas(engine, group=None, src: int = 0) -> None:
    """Start of a data-parallel run: every replica takes rank `src`'s parameters — the flat buffer (which also holds
    batch-norm running statistics) AND the parameters of user-defined modules that live outside it
    (set_encoder / set_decoder / set_classifier: engine._enc_params).

    Batch normalisation: per-shard batch statistics are NOT the global batch's, so a sharded step would differ from the
    single-process step the parity bar is defined on, and the running estimates would drift apart between replicas.
    Rejected rather than silently different (the reference has no distributed mode to mirror)."""
    if not (td.is_available() and td.is_initialized() and td.get_world_size(group) > 1):
        return
    if getattr(engine, "_bn_enc", None) or getattr(engine, "_bn_dec", None):
        raise NotImplementedError("batchnorm=True models are not trained data-parallel: per-shard batch statistics "
                                  "differ from the global batch's (train them on one GPU, or build with batchnorm=False)")
    broadcast_(engine.flat, src, group)
    for q in getattr(engine, "_enc_params", None) or []:
        td.broadcast(q.data, src=src, group=group)


# ---------------------------------------------------------------------------------------------------------------------------
# The collective inside the library (ABI v16: pv_dist_*, pv_ivae_dp_step, pv_ved_dp_step; csrc/pv_dist.hip).  torch.distributed's
# all_reduce runs on ProcessGroupNCCL's own stream between two event hand-offs; a NativeComm is an RCCL communicator created
# through ctypes on the RCCL library this process ALREADY holds (PyTorch's), whose ncclAllReduce the library then enqueues on the
# compute stream between the last gradient launch and the optimizer launch — one library call per step, nothing else on the way.
# torch.distributed is still what bootstraps it (the 128-byte ncclUniqueId travels over the existing process group).

class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def _loaded_rccl_path() -> Optional[str]:
    """Path of the RCCL shared object mapped into this process (torch's own copy once torch.cuda / torch.distributed is in), else
    torch's bundled one, else ROCm's."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "librccl" in os.path.basename(path):
                    return path
    except OSError:
        pass
    cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"]
    for c in cand:
        if os.path.exists(c):
            return c
    return None


class NativeComm:
    """An RCCL communicator over the ranks of `group` (default WORLD; works at world size 1 without a process group), owned by
    this object, usable by pv_dist_allreduce_sum / pv_ivae_dp_step / pv_ved_dp_step (`.handle`).  One per (process, device)."""

    def __init__(self, device=None, group=None):
        from . import _abi
        if not torch.cuda.is_available():
            raise _abi.PvError("pyroved_amd: NativeComm needs a HIP device (the collective is RCCL; there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.group = group
        ini = td.is_available() and td.is_initialized()
        self.rank, self.world = (td.get_rank(group), td.get_world_size(group)) if ini else (0, 1)
        path = _loaded_rccl_path()
        if path is None:
            raise _abi.PvError("pyroved_amd: no RCCL library found (librccl.so)")
        self._nccl = C.CDLL(path)
        _abi.check(_abi.lib().pv_dist_load(path.encode()), "pv_dist_load(%s)" % path)
        self.library = _abi.lib().pv_dist_library().decode()
        n = self._nccl
        n.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        n.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        n.ncclCommDestroy.argtypes = [C.c_void_p]
        n.ncclGetErrorString.restype = C.c_char_p
        n.ncclGetErrorString.argtypes = [C.c_int]
        uid = _NcclUniqueId()
        if self.rank == 0:
            self._ok(n.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            box = [bytes(bytearray(uid.internal))] if self.rank == 0 else [None]
            src = td.get_global_rank(group, 0) if group is not None else 0
            td.broadcast_object_list(box, src=src, group=group)
            C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        with torch.cuda.device(self.device):          # ncclCommInitRank binds the communicator to the CURRENT device
            self._ok(n.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.handle = comm
        r, w = C.c_int32(-1), C.c_int32(-1)
        _abi.check(_abi.lib().pv_dist_comm_info(self.handle, C.byref(r), C.byref(w)), "pv_dist_comm_info")
        if (r.value, w.value) != (self.rank, self.world):
            raise _abi.PvError("pyroved_amd: RCCL communicator is rank %d of %d, the process group says %d of %d"
                               % (r.value, w.value, self.rank, self.world))

    def _ok(self, rc, what):
        if rc != 0:
            from . import _abi
            raise _abi.PvError("pyroved_amd: %s failed: %s (%d)" % (what, self._nccl.ncclGetErrorString(rc).decode(), rc))

    def allreduce_sum_(self, flat: torch.Tensor) -> None:
        """In-place SUM over the communicator's ranks on the CURRENT stream of the tensor's device (pv_dist_allreduce_sum)."""
        from . import _abi
        _abi.require_device(flat, "flat")
        if not flat.is_contiguous():
            raise ValueError("allreduce_sum_: contiguous tensor expected")
        with _abi.device_of(flat.device):
            _abi.check(_abi.lib().pv_dist_allreduce_sum(self.handle, _abi.ptr(flat), flat.numel(), _abi.current_stream()),
                       "pv_dist_allreduce_sum")

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            torch.cuda.synchronize(self.device)
            self._nccl.ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        # (interpreter shutdown: the HIP runtime / RCCL may already be gone, and `import` no longer works — module-level sys)
        try:
            if _sys is None or _sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass


_native = {}


def native_comm(device, group=None) -> "NativeComm":
    """The process's NativeComm for (device, group), created on first use — a collective call: every rank of `group` must make it."""
    key = (torch.device(device).index, id(group) if group is not None else None)
    if key not in _native:
        _native[key] = NativeComm(device, group)
    return _native[key]


def native_available(group=None) -> bool:
    """Whether the data-parallel step can keep its collective in the library: a HIP device and — when a process group exists —
    the nccl (= RCCL) backend (gloo groups, the CPU tests' and the one-GPU test hooks', stay on torch.distributed)."""
    if not torch.cuda.is_available():
        return False
    if td.is_available() and td.is_initialized():
        return td.get_backend(group) == "nccl"
    return True


def close_native() -> None:
    """Destroys every NativeComm this process created through native_comm() (call before torch.distributed.destroy_process_group)."""
    for c in list(_native.values()):
        try:
            c.close()
        except Exception:
            pass
    _native.clear()
