"""RCCL communicator of libkartohip (kh_comm_*, include/karto_hip.h): one process per GPU, rank 0 makes the 128-byte
id and hands it to the other ranks by whatever the launcher offers (a torch.distributed broadcast, MPI, a file)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

ID_BYTES = 128


def unique_id() -> np.ndarray:
    out = np.zeros(ID_BYTES, dtype=np.uint8)
    capi.check(capi.lib().kh_comm_unique_id(out), "kh_comm_unique_id")
    return out


class Communicator:
    def __init__(self, device: int, rank: int, world: int, comm_id):
        self._h = C.c_void_p()
        comm_id = np.ascontiguousarray(comm_id, dtype=np.uint8)
        assert comm_id.shape == (ID_BYTES,)
        capi.check(capi.lib().kh_comm_create(device, rank, world, comm_id, C.byref(self._h)), "kh_comm_create")
        self.device, self.rank, self.world = device, rank, world

    @property
    def handle(self):
        return self._h

    def all_reduce_sum_f64(self, device_ptr: int, count: int, stream: int = 0):
        capi.check(capi.lib().kh_comm_allreduce_sum_f64(self._h, C.c_void_p(device_ptr), count, C.c_void_p(stream)),
                   "kh_comm_allreduce_sum_f64")

    def all_gather_f64(self, send_ptr: int, recv_ptr: int, count_per_rank: int, stream: int = 0):
        capi.check(capi.lib().kh_comm_allgather_f64(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), count_per_rank,
                                                    C.c_void_p(stream)), "kh_comm_allgather_f64")

    def close(self):
        if self._h:
            capi.lib().kh_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """n doubles in HBM (kh_device_*): what the collectives read and write."""

    def __init__(self, n: int, device: int = 0):
        self.n = int(n)
        self._p = C.c_void_p()
        capi.check(capi.lib().kh_device_malloc(device, 8 * self.n, C.byref(self._p)), "kh_device_malloc")

    @property
    def ptr(self) -> int:
        return self._p.value

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        capi.check(capi.lib().kh_device_upload(self._p, a.ctypes.data_as(C.c_void_p), 8 * self.n), "kh_device_upload")

    def download(self) -> np.ndarray:
        out = np.zeros(self.n)
        capi.check(capi.lib().kh_device_download(out.ctypes.data_as(C.c_void_p), self._p, 8 * self.n), "kh_device_download")
        return out

    def free(self):
        if self._p:
            capi.lib().kh_device_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
