// Integration test for RustAudio/lewton 0.10.2 with integration/lewton_tap_hashes/lewton_taps.patch applied (copy this file
// to <lewton>/tests/tap_hashes.rs): decodes the three fixture files of lewton_amd's tests/golden/ with lewton itself and
// prints, per file, the SHA-256 of the interleaved i16 PCM and of the four debug taps -- the very quantities
// tests/golden/tap_hashes.json holds for the CPU oracle of lewton_amd.
//
//     LEWTON_AMD_GOLDEN=/path/to/lewton_amd/tests/golden cargo test --test tap_hashes -- --nocapture --test-threads 1 \
//         | python /path/to/lewton_amd/tests/golden/check_tap_hashes.py
//
// No dependency beyond lewton and its `ogg` feature: the SHA-256 below is the FIPS 180-4 algorithm written out.
extern crate lewton;

use lewton::inside_ogg::OggStreamReader;
use std::fs::File;

const K :[u32; 64] = [
	0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
	0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
	0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
	0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
	0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
	0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
	0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
	0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
];

struct Sha256 {
	h :[u32; 8],
	buf :Vec<u8>,
	len :u64,
}

impl Sha256 {
	fn new() -> Self {
		Sha256 {
			h : [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19],
			buf : Vec::new(),
			len : 0,
		}
	}
	fn block(&mut self, b :&[u8]) {
		let mut w = [0u32; 64];
		for i in 0 .. 16 {
			w[i] = ((b[4 * i] as u32) << 24) | ((b[4 * i + 1] as u32) << 16) | ((b[4 * i + 2] as u32) << 8) | (b[4 * i + 3] as u32);
		}
		for i in 16 .. 64 {
			let s0 = w[i - 15].rotate_right(7) ^ w[i - 15].rotate_right(18) ^ (w[i - 15] >> 3);
			let s1 = w[i - 2].rotate_right(17) ^ w[i - 2].rotate_right(19) ^ (w[i - 2] >> 10);
			w[i] = w[i - 16].wrapping_add(s0).wrapping_add(w[i - 7]).wrapping_add(s1);
		}
		let mut v = self.h;
		for i in 0 .. 64 {
			let s1 = v[4].rotate_right(6) ^ v[4].rotate_right(11) ^ v[4].rotate_right(25);
			let ch = (v[4] & v[5]) ^ (!v[4] & v[6]);
			let t1 = v[7].wrapping_add(s1).wrapping_add(ch).wrapping_add(K[i]).wrapping_add(w[i]);
			let s0 = v[0].rotate_right(2) ^ v[0].rotate_right(13) ^ v[0].rotate_right(22);
			let maj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
			let t2 = s0.wrapping_add(maj);
			v[7] = v[6];
			v[6] = v[5];
			v[5] = v[4];
			v[4] = v[3].wrapping_add(t1);
			v[3] = v[2];
			v[2] = v[1];
			v[1] = v[0];
			v[0] = t1.wrapping_add(t2);
		}
		for i in 0 .. 8 {
			self.h[i] = self.h[i].wrapping_add(v[i]);
		}
	}
	fn update(&mut self, data :&[u8]) {
		self.len += data.len() as u64;
		self.buf.extend_from_slice(data);
		let full = self.buf.len() / 64 * 64;
		let rest = self.buf.split_off(full);
		let blocks = ::std::mem::replace(&mut self.buf, rest);
		for b in blocks.chunks(64) {
			self.block(b);
		}
	}
	fn hex(mut self) -> String {
		let bits = self.len.wrapping_mul(8);
		let mut tail = ::std::mem::replace(&mut self.buf, Vec::new());
		tail.push(0x80);
		while tail.len() % 64 != 56 {
			tail.push(0);
		}
		for i in 0 .. 8 {
			tail.push((bits >> (56 - 8 * i)) as u8);
		}
		for b in tail.chunks(64) {
			self.block(b);
		}
		let mut s = String::new();
		for x in self.h.iter() {
			s.push_str(&format!("{:08x}", x));
		}
		s
	}
}

#[test]
fn sha256_known_answer() {
	let mut h = Sha256::new();
	h.update(b"abc");
	assert_eq!(h.hex(), "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad");
	let mut h = Sha256::new();
	h.update(b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq");
	assert_eq!(h.hex(), "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1");
}

fn hashes_of(dir :&str, name :&str) {
	let f = File::open(format!("{}/{}", dir, name)).expect("fixture file");
	for k in 0 .. 4 {
		lewton::tap::take(k);
	}
	let mut srr = OggStreamReader::new(f).expect("headers");
	let mut pcm = Sha256::new();
	let mut taps = [Sha256::new(), Sha256::new(), Sha256::new(), Sha256::new()];
	let mut tap_values = [0usize; 4];
	let mut values = 0usize;
	let mut packets = 0usize;
	while let Some(samples) = srr.read_dec_packet_itl().expect("decode") {
		let mut bytes = Vec::with_capacity(samples.len() * 2);
		for s in samples.iter() {
			bytes.extend_from_slice(&s.to_le_bytes());
		}
		pcm.update(&bytes);
		values += samples.len();
		packets += 1;
		for k in 0 .. 4 {
			let t = lewton::tap::take(k);
			tap_values[k] += t.len() / 4;
			taps[k].update(&t);
		}
	}
	println!("TAPHASH {} audio_packets {}", name, packets);
	println!("TAPHASH {} pcm_i16_interleaved {} {}", name, pcm.hex(), values);
	let names = ["residue_pre_inverse", "residue_post_inverse", "pre_mdct", "post_mdct"];
	let [t0, t1, t2, t3] = taps;
	for (k, t) in vec![t0, t1, t2, t3].into_iter().enumerate() {
		println!("TAPHASH {} {} {} {}", name, names[k], t.hex(), tap_values[k]);
	}
}

#[test]
fn tap_hashes() {
	let dir = ::std::env::var("LEWTON_AMD_GOLDEN").expect("set LEWTON_AMD_GOLDEN to lewton_amd's tests/golden directory");
	for name in ["invalid_keypress.ogg", "synth_stereo_mixed.ogg", "synth_surround51.ogg"].iter() {
		hashes_of(&dir, name);
	}
}
