"""Staging ring (lw_ring_*): entropy decode of batch N+1 on the host threads overlaps H2D, synthesis kernels and D2H of
batch N (BASELINE north_star).  Slots are FIFO; a PreviousWindowRight sees its packets in submission order.

    ring = Ring(decoder, slots=3, max_packets=4096, samples='i16')
    ring.submit(marshalled)            # returns while the GPU works
    res, pcm = ring.collect()          # oldest batch: [(status, n_samples, out_offset)], numpy view of the pinned PCM
    ring.release()
"""
import ctypes as C

import numpy as np

from . import _native as N
from .audio import _FMT


class Ring:
    def __init__(self, decoder, slots=3, max_packets=4096, samples="i16"):
        err = C.c_int(0)
        self.dec = decoder
        self.fmt = _FMT[samples]
        self.max_packets = max_packets
        self._h = N.lw_ring_create(decoder._h, slots, max_packets, self.fmt, C.byref(err))
        if not self._h:
            raise RuntimeError("lw_ring_create failed (%d): %s" % (err.value, N.device_error()))

    def close(self):
        if getattr(self, "_h", None):
            N.lw_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        if N is not None and getattr(N, "lw_ring_destroy", None) is not None:
            self.close()

    def marshal(self, packets):
        """lw_packet array for `packets` (list of (bytes, PreviousWindowRight)); build once, submit many times."""
        n = len(packets)
        arr = (N.Packet * n)()
        bufs = []
        for i, (data, pwr) in enumerate(packets):
            data = bytes(data)
            bufs.append(data)
            arr[i].data = C.cast(C.c_char_p(data), C.c_void_p)
            arr[i].len = len(data)
            arr[i].pwr = pwr._bind(self.dec)
        return (arr, bufs, n)

    def set_entropy_on_device(self, on=True):
        """Entropy stage on the device (lw_ring_set_entropy_on_device): the packets themselves cross PCIe, k_entropy decodes
        floors and residues, one lane per packet.  Returns False (host stage stays) when the stream is not eligible."""
        rc = N.lw_ring_set_entropy_on_device(self._h, 1 if on else 0)
        if rc == N.ERR_UNSUPPORTED:
            return False
        if rc:
            raise RuntimeError("lw_ring_set_entropy_on_device: %d" % rc)
        return True

    def _call(self, name, rc):
        if rc:
            raise RuntimeError("%s: %d %s" % (name, rc, N.device_error()))

    def stage(self, marshalled, n_threads=0):
        # lw_ring_stage is synchronous: every packet has been decoded (or copied into pinned staging) when it returns, so
        # `marshalled` only has to live for the duration of this call -- the local reference does that
        arr, bufs, n = marshalled
        self._call("lw_ring_stage", N.lw_ring_stage(self._h, arr, n, n_threads))

    def launch(self):
        self._call("lw_ring_launch", N.lw_ring_launch(self._h))

    def submit(self, marshalled, n_threads=0):
        arr, bufs, n = marshalled
        self._call("lw_ring_submit", N.lw_ring_submit(self._h, arr, n, n_threads))

    @property
    def in_flight(self):
        return N.lw_ring_in_flight(self._h)

    @property
    def slots(self):
        return N.lw_ring_slots(self._h)

    @property
    def last_kernels(self):
        return (N.lw_ring_last_kernels(self._h) or b"").decode()

    def collect(self, copy=True):
        res = C.POINTER(N.PacketResult)()
        n, pcm, elems = C.c_size_t(0), C.c_void_p(0), C.c_size_t(0)
        self._call("lw_ring_collect", N.lw_ring_collect(self._h, C.byref(res), C.byref(n), C.byref(pcm), C.byref(elems)))
        out = [(res[i].status, res[i].n_samples, res[i].out_offset) for i in range(n.value)]
        dt = np.float32 if self.fmt == N.FMT_F32_PLANAR else np.int16
        if elems.value:
            buf = (C.c_char * (elems.value * np.dtype(dt).itemsize)).from_address(pcm.value)
            a = np.frombuffer(buf, dtype=dt)
            a = a.copy() if copy else a
        else:
            a = np.zeros(0, dt)
        return out, a

    def collect_nocopy(self):
        """Wait for the oldest batch; returns (n_packets, pcm_elems) without touching the data (throughput loops)."""
        n, elems = C.c_size_t(0), C.c_size_t(0)
        self._call("lw_ring_collect", N.lw_ring_collect(self._h, None, C.byref(n), None, C.byref(elems)))
        return n.value, elems.value

    def release(self):
        self._call("lw_ring_release", N.lw_ring_release(self._h))

    def drain(self):
        self._call("lw_ring_drain", N.lw_ring_drain(self._h))
