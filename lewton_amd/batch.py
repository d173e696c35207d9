"""Batched decode: many packets (of one or many streams) per launch -- where the throughput comes from.

    dec = Decoder(ident, setup, device)
    b = Batch(dec, max_packets, samples='i16')
    res = b.entropy([(packet_bytes, pwr), ...])      # host stage -> pinned staging
    b.upload(stream)                                  # hipMemcpyAsync
    b.synth(device_ptr, capacity_elems, stream)       # HIP kernels (async)   or   b.synth_to_host()
"""
import ctypes as C

import numpy as np

from . import _native as N
from .audio import _FMT, Decoder, PreviousWindowRight  # noqa: F401


class Batch:
    def __init__(self, decoder, max_packets, samples="i16"):
        err = C.c_int(0)
        self.dec = decoder
        self.fmt = _FMT[samples]
        self.max_packets = max_packets
        self._h = N.lw_batch_create(decoder._h, max_packets, self.fmt, C.byref(err))
        if not self._h:
            raise RuntimeError("lw_batch_create failed (%d): %s" % (err.value, N.device_error()))
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            N.lw_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        if N is not None and getattr(N, "lw_batch_destroy", None) is not None:  # not during interpreter shutdown
            self.close()

    def marshal(self, packets):
        """Build the lw_packet array for `packets` once (list of (bytes, PreviousWindowRight)); reusable with entropy_marshalled."""
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

    def entropy_marshalled(self, marshalled, n_threads=0):
        arr, bufs, n = marshalled
        self._keep = (arr, bufs)
        rc = N.lw_batch_entropy(self._h, arr, n, n_threads)
        if rc:
            raise RuntimeError("lw_batch_entropy: %d" % rc)

    def entropy(self, packets, n_threads=0):
        """packets: list of (bytes, PreviousWindowRight).  Returns the per-packet results array."""
        self.entropy_marshalled(self.marshal(packets), n_threads)
        return self.results()

    def results(self):
        n = N.lw_batch_size(self._h)
        p = N.lw_batch_results(self._h)
        return [(p[i].status, p[i].n_samples, p[i].out_offset) for i in range(n)]

    @property
    def out_elems(self):
        return N.lw_batch_out_elems(self._h)

    @property
    def algorithmic_bytes(self):
        return N.lw_batch_algorithmic_bytes(self._h)

    @property
    def state_bytes(self):
        """window state that crosses HBM at the launch boundary (not part of algorithmic_bytes; lw_batch_state_bytes)"""
        return N.lw_batch_state_bytes(self._h)

    @property
    def last_kernels(self):
        return (N.lw_batch_last_kernels(self._h) or b"").decode()

    def set_force_generic(self, on):
        N.lw_batch_set_force_generic(self._h, 1 if on else 0)

    def debug_set_mix(self, mode):
        """test hook (lw_debug_batch_set_mix): 0 = two launches for a mixed short / long batch, -1 = k_mix where it applies"""
        N.lw_debug_batch_set_mix(self._h, int(mode))

    def debug_set_long10(self, mode):
        """test hook (lw_debug_batch_set_long10) for blocksize_1 = 10 / 12 streams: -1 = k_long10 / k_long12 incl. their EDGE form,
        1 = without the EDGE form (long blocks next to short ones through the generic kernels), 0 = k_short<32> / k_big<12>"""
        N.lw_debug_batch_set_long10(self._h, int(mode))

    def debug_break_mix(self, spin):
        """test hook (lw_debug_batch_break_mix): the long blocks' waves of k_mix never signal; the short blocks' waves give up after
        `spin` polls and the batch fails with LW_ERR_DEVICE (0 = normal operation)"""
        N.lw_debug_batch_break_mix(self._h, int(spin))

    def device_status(self):
        """lw_batch_device_status: 0, or LW_ERR_DEVICE when a kernel of the completed launches raised the batch's error word"""
        return N.lw_batch_device_status(self._h)

    def debug_set_halo(self, mode):
        """test hook (lw_debug_batch_set_halo): 0 = predecessors of chunk starts by the pre-pass launch, -1 = inside the launch (default)"""
        N.lw_debug_batch_set_halo(self._h, int(mode))

    def debug_set_rounds(self, rounds):
        """test hook (lw_debug_batch_set_rounds): rounds per workgroup of the specialised kernel, 0 = planner's choice"""
        N.lw_debug_batch_set_rounds(self._h, int(rounds))

    def upload(self, stream=None):
        rc = N.lw_batch_upload(self._h, stream)
        if rc:
            raise RuntimeError("lw_batch_upload: %d %s" % (rc, N.device_error()))

    def synth(self, device_ptr, capacity_elems, stream=None):
        rc = N.lw_batch_synth(self._h, device_ptr, capacity_elems, stream)
        if rc:
            raise RuntimeError("lw_batch_synth: %d %s" % (rc, N.device_error()))

    def synth_to_host(self, stream=None):
        n = self.out_elems
        out = np.zeros(max(n, 1), np.float32 if self.fmt == N.FMT_F32_PLANAR else np.int16)
        rc = N.lw_batch_synth_to_host(self._h, out.ctypes.data_as(C.c_void_p), n, stream)
        if rc:
            raise RuntimeError("lw_batch_synth_to_host: %d %s" % (rc, N.device_error()))
        return out[:n]

    def tap(self, idx, which, ch, n):
        want = ch * n if which == N.TAP_POST_MDCT else ch * n // 2
        out = np.zeros(want, np.float32)
        rc = N.lw_batch_tap(self._h, idx, which, out.ctypes.data_as(N.f32p), want)
        if rc:
            raise RuntimeError("lw_batch_tap: %d %s" % (rc, N.device_error()))
        return out.reshape(ch, -1)

    def set_entropy_on_device(self, on=True):
        """Entropy stage on the device (lw_batch_set_entropy_on_device, k_entropy).  Returns False when the stream is not
        eligible (lw_decoder_supports_device_entropy says why)."""
        rc = N.lw_batch_set_entropy_on_device(self._h, 1 if on else 0)
        if rc == N.ERR_UNSUPPORTED:
            return False
        if rc:
            raise RuntimeError("lw_batch_set_entropy_on_device: %d %s" % (rc, N.device_error()))
        return True

    def split(self, flat, channels):
        """Split the flat output of synth_to_host into per-packet arrays ([ch][m], or [m*ch] interleaved)."""
        out = []
        for status, m, off in self.results():
            if status != 0:
                out.append(None)
            elif self.fmt == N.FMT_I16_INTERLEAVED:
                out.append(flat[off: off + m * channels])
            else:
                out.append(flat[off: off + m * channels].reshape(channels, m))
        return out
