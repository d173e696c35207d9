// Host side of the device entropy stage (product code): flattens a setup header into the image lw_dev_entropy.h decodes
// from, and says whether a stream is eligible.
#pragma once

#include "lw_dev_entropy.h"
#include "lw_host.hpp"

#include <vector>

namespace lw {

struct DevEntropyImage {
	std::vector<uint8_t> blob; // [books][floors][residues][modes][lut u32][vq f32][digits u16][runs][tree nodes i32], sections 16-byte aligned
	size_t off_books = 0, off_floors = 0, off_residues = 0, off_modes = 0, off_lut = 0, off_vq = 0, off_digits = 0, off_runs = 0, off_nodes = 0;
	uint32_t ch = 0, fstride = 0, ws_bytes = 0, res_floats = 0, general = 0;
};

// false + *why when the stream has to stay on the host entropy stage
bool dev_entropy_build(const Ident &id, const Setup &s, unsigned fstride, DevEntropyImage &img, const char **why);

// the image seen through pointers into `base` (the blob itself on the host, its copy in HBM on the device)
LwEntTables dev_entropy_view(const DevEntropyImage &img, const uint8_t *base);

} // namespace lw
