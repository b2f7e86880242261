"""The path's collective through the C ABI (include/dissc_hip.h: dissc_comm_*, dissc_allgather_waves) -- the route a
maintainer takes who binds libdissc_hip.so without torch.distributed.  The CLIs keep torch.distributed (backend "nccl" = the
same RCCL) as their default; ``DISSC_COLLECTIVE=cabi`` makes the harness gather through a WaveComm instead.

Reference: sr/inference.py:288-292,351-354 (Pool(8): every worker hands its results back) -> one all-gather of the packed
waveform buffers (dissc_pack_rows' layout) per round.

A WaveComm quacks like the slice of ``torch.distributed`` the harness uses (all_gather_into_tensor), so
``harness.gather_store(..., dist=WaveComm(...))`` runs the same round on it."""
import ctypes

import torch

from . import _lib

ID_BYTES = 128  # DISSC_COMM_ID_BYTES


def unique_id():
    """bytes(128): made by ONE rank (ncclGetUniqueId) and handed to the others out of band (file, socket, store)."""
    buf = ctypes.create_string_buffer(ID_BYTES)
    _lib.check(_lib.lib.dissc_comm_unique_id(buf), "dissc_comm_unique_id")
    return buf.raw


class _Done:
    """what torch's async_op=True returns, for a collective that is simply enqueued on a stream"""
    def __init__(self, stream):
        self.stream = stream

    def wait(self):  # the caller's current stream waits for the collective's stream (no host block), like Work.wait() on NCCL
        torch.cuda.current_stream().wait_stream(self.stream)
        return True


class WaveComm:
    """An RCCL communicator made and used through the C ABI.  Collective constructor: every rank calls it with the same id,
    each with its own GPU current (RCCL: one GPU per rank)."""

    def __init__(self, uid, nranks, rank, device=None):
        if len(uid) != ID_BYTES:
            raise ValueError("uid must be the 128 bytes unique_id() returned on one rank")
        self.nranks, self.rank = int(nranks), int(rank)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.dissc_comm_create(uid, self.nranks, self.rank, ctypes.byref(h)), "dissc_comm_create")
        self._h = h

    def get_world_size(self):
        return self.nranks

    def get_rank(self):
        return self.rank

    def all_gather_into_tensor(self, out, buf, async_op=False, stream=None):
        """out [nranks * n] <- every rank's buf [n] (fp32, contiguous, on this communicator's device), on ``stream`` (default: the
        current stream).  Like torch.distributed on NCCL the call only enqueues; async_op=True returns an object with wait()."""
        if self._h is None:
            raise _lib.DisscError("WaveComm: destroyed")
        if buf.dtype != torch.float32 or out.dtype != torch.float32 or not (buf.is_contiguous() and out.is_contiguous()):
            raise ValueError("all_gather_into_tensor: contiguous float32 tensors")
        if out.numel() != self.nranks * buf.numel() or buf.device != self.device or out.device != self.device:
            raise ValueError("all_gather_into_tensor: out must hold nranks * buf.numel() floats on the communicator's device")
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.dissc_allgather_waves(self._h, buf.data_ptr(), buf.numel(), out.data_ptr(),
                                                      ctypes.c_void_p(st.cuda_stream)), "dissc_allgather_waves")
        return _Done(st) if async_op else None

    def destroy(self):
        if self._h is not None:
            torch.cuda.synchronize(self.device)
            h, self._h = self._h, None
            _lib.check(_lib.lib.dissc_comm_destroy(h), "dissc_comm_destroy")

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass
