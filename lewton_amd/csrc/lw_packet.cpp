// The drop-in single-packet call of the MI355X decode path (product code): audio::read_audio_packet_generic
// (src/audio.rs:919) as one synchronous host -> device -> host round trip through an internal one-packet batch.
#include "lw_internal.hpp"

#include <algorithm>
#include <cstring>

extern "C" {

// ---- one packet ---------------------------------------------------------------------------------
int lw_read_audio_packet(lw_decoder *d, const uint8_t *packet, size_t len, lw_pwr *pwr, int fmt, void *out,
		size_t cap_per_channel, size_t *n_samples)
{
	if (!d || (!packet && len) || !pwr || !out || !n_samples)
		return LW_ERR_NULL_ARG;
	if (pwr->dec != d)
		return LW_ERR_STATE_MISMATCH;
	if (fmt < 0 || fmt > 2)
		return LW_ERR_NULL_ARG;
	if (int rc = lw_decoder_set_device(d))
		return rc;
	if (!d->one || d->one->fmt != fmt) {
		if (d->one)
			lw_batch_destroy(d->one);
		int e = 0;
		d->one = lw_batch_create(d, 1, fmt, &e);
		if (!d->one)
			return e ? e : LW_ERR_DEVICE;
	}
	lw_batch *b = d->one;
	lw_packet pk{packet, len, pwr};
	// lw_batch_entropy commits the host half of the PreviousWindowRight (present, len, parity) when it plans the batch; the
	// device half follows when the kernels run.  Any failure in between must leave `pwr` as the reference leaves it on an
	// error: untouched (the one error that consumes the state, audio.rs:1107-1111, is reported through res.status).
	const lw_pwr saved = *pwr;
	if (int rc = lw_batch_entropy(b, &pk, 1, 1)) {
		*pwr = saved;
		return rc;
	}
	const lw_packet_result &res = b->results[0];
	if (res.status != LW_OK)
		return res.status;
	int rc = LW_OK;
	if (res.n_samples > cap_per_channel)
		rc = LW_AUDIO_BUFFER_NOT_ADDRESSABLE;
	if (!rc)
		rc = lw_batch_upload(b, nullptr);
	const size_t need = std::max<size_t>(b->out_elems, 1) * lw_elem_size(fmt);
	if (!rc && d->one_out_bytes < need) {
		if (d->one_out)
			(void)hipHostFree(d->one_out);
		d->one_out = nullptr;
		d->one_out_bytes = 0;
		const size_t bytes = std::max<size_t>(need, (size_t)d->T.state_stride * 2 * 4);
		if (lw_hip_ok(hipHostMalloc(&d->one_out, bytes), "hipHostMalloc(packet output)"))
			d->one_out_bytes = bytes;
		else
			rc = LW_ERR_DEVICE;
	}
	// the kernels write the PCM straight into the pinned host buffer (device-visible): launch + synchronise, no D2H copy
	if (!rc)
		rc = lw_batch_synth(b, d->one_out, b->out_elems, nullptr); // (a first packet yields no samples, only the state)
	if (!rc && !lw_hip_ok(hipStreamSynchronize(nullptr), "hipStreamSynchronize"))
		rc = LW_ERR_DEVICE;
	if (rc) {
		*pwr = saved; // the packet was not decoded: the next call overlaps against the state this one found
		return rc;
	}
	std::memcpy(out, d->one_out, b->out_elems * lw_elem_size(fmt));
	*n_samples = res.n_samples;
	return LW_OK;
}

} // extern "C"
