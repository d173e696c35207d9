"""Sharding of independent streams across the GPUs of a node: whole streams, `stream_id mod G`.

A stream's decode touches only its own PreviousWindowRight and the immutable headers (audio.rs:919), so streams never
exchange data: one process per GPU, one `lw_decoder` per process, no collective in the data path.  The only
communication is the benchmark's barrier and the MAX-reduction of the elapsed time.
"""


def shard_streams(n_streams, world_size, rank):
    """Stream ids owned by `rank`: stream_id mod world_size == rank."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, n_streams, world_size))


def owner_of(stream_id, world_size):
    return stream_id % world_size


def max_elapsed(elapsed_seconds, dist=None, device=None):
    """Whole-job time = the slowest rank's time (dist: an initialised torch.distributed module, or None for 1 process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_seconds)
    import torch
    t = torch.tensor([elapsed_seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank, elapsed_seconds, dist=None, device=None):
    """Weak scaling: every rank processed `units_per_rank` units; value = all units / slowest rank's time."""
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    return units_per_rank * world / max_elapsed(elapsed_seconds, dist, device)


class Sharder:
    """Single-process form (lw_sharder_*): one decoder, staging ring and worker thread per entry of `devices`; streams are
    opened through the sharder and live on shard stream_id mod G.  decode() is synchronous; submit() / collect() keep up to
    three calls in flight so that every shard's host stage overlaps the GPU work of the call before."""

    def __init__(self, ident, setup, devices, max_packets_per_shard, samples="i16", share_cus=False):
        """share_cus: logical shards (a device named several times) each launch on their own CUs of every XCD
        (lw_decoder_set_cu_share; measurement hook, tools/probe/sharder_probe.py); False = all of them on the whole device"""
        import ctypes as C
        from . import _native as N
        from .audio import _FMT
        self._N, self._C = N, C
        self.ident, self.fmt = ident, _FMT[samples]
        arr = (C.c_int * len(devices))(*devices)
        err = C.c_int(0)
        N.lw_debug_sharder_share_cus(1 if share_cus else 0)
        self._h = N.lw_sharder_create(ident._h, setup._h, arr, len(devices), max_packets_per_shard, self.fmt, C.byref(err))
        N.lw_debug_sharder_share_cus(0)
        if not self._h:
            raise RuntimeError("lw_sharder_create failed (%d): %s" % (err.value, N.device_error()))
        self._streams = {}

    @property
    def shards(self):
        return self._N.lw_sharder_shards(self._h)

    def shard_cus(self, shard):
        """compute units the shard's launches run on"""
        return self._N.lw_sharder_shard_cus(self._h, shard)

    def shard_of(self, stream_id):
        return self._N.lw_sharder_shard_of(self._h, stream_id)

    def set_entropy_on_device(self, on=True):
        """every shard decodes floors and residues on its own GPU (k_entropy); False when the stream is not eligible"""
        rc = self._N.lw_sharder_set_entropy_on_device(self._h, 1 if on else 0)
        if rc == self._N.ERR_UNSUPPORTED:
            return False
        if rc:
            raise RuntimeError("lw_sharder_set_entropy_on_device: %d" % rc)
        return True

    def stream(self, stream_id):
        if stream_id not in self._streams:
            h = self._N.lw_sharder_stream_open(self._h, stream_id)
            if not h:
                raise RuntimeError("lw_sharder_stream_open failed: " + self._N.device_error())
            self._streams[stream_id] = h
        return self._streams[stream_id]

    def decode(self, packets, n_threads=0):
        """packets: list of (stream_id, bytes).  Returns a list of per-packet arrays ([ch][m], None for a failed packet) and
        the list of (status, n_samples, out_offset)."""
        import numpy as np
        N, C = self._N, self._C
        n = len(packets)
        arr = (N.ShardPacket * n)()
        keep = []
        for i, (sid, data) in enumerate(packets):
            data = bytes(data)
            keep.append(data)
            arr[i].stream = self.stream(sid)
            arr[i].data = C.cast(C.c_char_p(data), C.c_void_p)
            arr[i].len = len(data)
        ch = self.ident.audio_channels
        cap = n * ch * (1 << self.ident.blocksize_1)
        out = np.zeros(cap, np.float32 if self.fmt == N.FMT_F32_PLANAR else np.int16)
        res = (N.PacketResult * n)()
        rc = N.lw_sharder_decode(self._h, arr, n, n_threads, out.ctypes.data_as(C.c_void_p), cap, res)
        if rc:
            raise RuntimeError("lw_sharder_decode: %d %s" % (rc, N.device_error()))
        blocks = []
        for i in range(n):
            if res[i].status != 0:
                blocks.append(None)
            elif self.fmt == N.FMT_I16_INTERLEAVED:
                blocks.append(out[res[i].out_offset: res[i].out_offset + res[i].n_samples * ch])
            else:
                blocks.append(out[res[i].out_offset: res[i].out_offset + res[i].n_samples * ch].reshape(ch, res[i].n_samples))
        return blocks, [(res[i].status, res[i].n_samples, res[i].out_offset) for i in range(n)]

    def marshal(self, packets):
        """lw_shard_packet array for `packets` (list of (stream_id, bytes)); build once, submit many times."""
        N, C = self._N, self._C
        n = len(packets)
        arr = (N.ShardPacket * n)()
        keep = []
        for i, (sid, data) in enumerate(packets):
            data = bytes(data)
            keep.append(data)
            arr[i].stream = self.stream(sid)
            arr[i].data = C.cast(C.c_char_p(data), C.c_void_p)
            arr[i].len = len(data)
        return arr, keep, n

    def submit(self, marshalled, n_threads=0):
        """lw_sharder_submit: every shard stages and launches its part; returns the elements the call will produce."""
        arr, _keep, n = marshalled
        elems = self._C.c_size_t(0)
        before = self._N.lw_sharder_in_flight(self._h)
        rc = self._N.lw_sharder_submit(self._h, arr, n, n_threads, self._C.byref(elems))
        self._pending = getattr(self, "_pending", [])
        if rc:
            msg = "lw_sharder_submit: %d %s" % (rc, self._N.device_error())
            if self._N.lw_sharder_in_flight(self._h) > before:
                # a shard failed after others had launched: the call is queued so that its slots can be freed -- it is the
                # YOUNGEST call, so everything in front of it is collected (and dropped) with it
                self._pending.append((n, elems.value))
                while self._pending:
                    cn, _ce = self._pending.pop(0)
                    scratch = (self._N.PacketResult * max(1, cn))()
                    self._N.lw_sharder_collect(self._h, None, 0, scratch, cn)
                    if self._N.lw_sharder_in_flight(self._h) == len(self._pending) + 1:
                        break   # (collect refused: leave the rest to close())
            raise RuntimeError(msg)
        self._pending.append((n, elems.value))
        return elems.value

    @property
    def in_flight(self):
        return self._N.lw_sharder_in_flight(self._h)

    def collect(self, out=None, want_results=True):
        """lw_sharder_collect of the oldest call: (flat sample array, [(status, n_samples, out_offset)]).  `out`: a numpy
        array to receive the samples (reused by throughput loops), else a fresh one; want_results=False skips building the
        Python list (the C array is still filled and checked for the call's status)."""
        import numpy as np
        N, C = self._N, self._C
        n, elems = self._pending[0]
        if out is None or out.size < elems:
            out = np.zeros(max(1, elems), np.float32 if self.fmt == N.FMT_F32_PLANAR else np.int16)
        if getattr(self, "_res_cap", 0) < n:
            self._res = (N.PacketResult * max(1, n))()
            self._res_cap = n
        res = self._res
        before = N.lw_sharder_in_flight(self._h)
        rc = N.lw_sharder_collect(self._h, out.ctypes.data_as(C.c_void_p), out.size, res, n)
        if rc:
            # the argument checks (LW_ERR_CAPACITY / LW_ERR_NULL_ARG) consume nothing; a shard's error has consumed the call
            if N.lw_sharder_in_flight(self._h) < before:
                self._pending.pop(0)
            raise RuntimeError("lw_sharder_collect: %d %s" % (rc, N.device_error()))
        self._pending.pop(0)
        if not want_results:
            return out[:elems], None
        return out[:elems], [(res[i].status, res[i].n_samples, res[i].out_offset) for i in range(n)]

    def collect_pinned(self, want_results=True):
        """lw_sharder_collect_pinned: per shard a numpy VIEW of its pinned PCM (valid until release()), and the results with
        out_offset relative to the owning shard's block."""
        import numpy as np
        N, C = self._N, self._C
        n, _elems = self._pending[0]
        if getattr(self, "_res_cap", 0) < n:
            self._res = (N.PacketResult * max(1, n))()
            self._res_cap = n
        G = self.shards
        pcm = (C.c_void_p * G)()
        el = (C.c_size_t * G)()
        rc = N.lw_sharder_collect_pinned(self._h, self._res, n, pcm, el)
        if rc:
            # a shard's error leaves the call held by the caller (see lewton_amd.h): give it back so that the C queue and
            # self._pending stay in step.  lw_sharder_release RETIRES the call and returns that part's status (LW_ERR_DEVICE
            # again, not 0), so whether the call left the queue is read off lw_sharder_in_flight, as collect() does; after a
            # failed argument check nothing is held and the release retires nothing
            before = N.lw_sharder_in_flight(self._h)
            N.lw_sharder_release(self._h)
            if N.lw_sharder_in_flight(self._h) < before:
                self._pending.pop(0)
            raise RuntimeError("lw_sharder_collect_pinned: %d %s" % (rc, N.device_error()))
        dt = np.float32 if self.fmt == N.FMT_F32_PLANAR else np.int16
        views = []
        for g in range(G):
            if el[g]:
                buf = (C.c_char * (el[g] * np.dtype(dt).itemsize)).from_address(pcm[g])
                views.append(np.frombuffer(buf, dtype=dt))
            else:
                views.append(np.zeros(0, dt))
        res = [(self._res[i].status, self._res[i].n_samples, self._res[i].out_offset) for i in range(n)] if want_results else None
        return views, res

    def release(self):
        # (a shard's error is returned by the release that retires the call: the queues stay in step either way)
        before = self._N.lw_sharder_in_flight(self._h)
        rc = self._N.lw_sharder_release(self._h)
        if rc == 0 or self._N.lw_sharder_in_flight(self._h) < before:
            self._pending.pop(0)
        if rc:
            raise RuntimeError("lw_sharder_release: %d" % rc)

    def close(self):
        if getattr(self, "_h", None):
            for h in self._streams.values():
                self._N.lw_sharder_stream_close(h)
            self._streams = {}
            self._N.lw_sharder_destroy(self._h)
            self._h = None

    def __del__(self):
        if getattr(self, "_N", None) is not None and getattr(self._N, "lw_sharder_destroy", None) is not None:
            self.close()
