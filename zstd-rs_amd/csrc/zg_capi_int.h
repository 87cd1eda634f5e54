// zg_capi_int.h — what the translation units behind include/zgpu.h share: the context, the dictionaries, the FrameDecoder mirror's
// state. Internal (not installed): the C ABI sees these as opaque types.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "../../include/zgpu.h"
#include "zg_engine.h"
#include "zg_xxh64.h"

struct ZgDict {   // Dictionary (decoding/dictionary.rs:12-37), tables in the engine's packed formats
  uint32_t id = 0;
  std::vector<uint32_t> fse;   // one FSE arena slot
  uint8_t logs[4] = {0, 0, 0, 0};
  std::vector<uint16_t> huf;
  uint8_t huf_maxbits = 0;
  uint32_t hist[3] = {1, 4, 8};
  std::vector<uint8_t> content;
};
struct zgpu_ctx {
  zg::Engine* eng = nullptr;
  std::map<uint32_t, ZgDict> dicts;   // FrameDecoder::dicts (frame_decoder.rs:82)
  std::string err;
};

namespace zg { class StreamCore; }

// ---- FrameDecoder mirror (frame_decoder.rs:80-627) ---------------------------------------------------------------------------
struct zgpu_decoder {
  zgpu_ctx* ctx = nullptr;
  bool has_state = false;
  zg::FrameHeader fh;
  uint64_t window_size = 0;
  bool frame_finished = false;
  uint64_t block_counter = 0, bytes_read = 0;
  bool has_checksum = false;
  uint32_t checksum = 0;
  uint32_t using_dict = 0;
  zg::FrameState fs;             // device side of DecoderScratch
  std::vector<uint8_t> buf;      // decoded, not yet drained bytes (DecodeBuffer, decode_buffer.rs:9-17)
  size_t head = 0;
  zg::Xxh64 hash;
  bool hash_on = true;           // ruzstd's `hash` cargo feature (default on): XXH64 of the drained bytes (decode_buffer.rs:42,223-227)
  size_t held() const { return buf.size() - head; }
  uint32_t drain_rule = 0;       // how the surface driving this decoder drains the reference's DecodeBuffer inside one run (zg_exact.h: ZG_DRAIN_*)
  uint64_t read_ahead = 0;       // zgpu_decoder_set_read_ahead: decode_blocks(UptoBytes(n)) decodes at least this many bytes per call (0: exactly the reference's n)
  zg::StreamCore* stream = nullptr;   // the streaming decoder that drives this decoder (zg_stream.h): while it reads ahead, the counters and the
                                      // buffered bytes live there and the accessors below ask it
};

// (zg_capi.cpp) DecodeBuffer::drain_to (decode_buffer.rs:256-314) on the mirror's host buffer; DecoderScratch::init_from_dict (scratch.rs:70-78)
size_t zg_dec_drain(zgpu_decoder* d, size_t n, uint8_t* dst);
int zg_apply_dict(zgpu_decoder* d, const ZgDict& dict);

// (zg_stream.cpp) what the decoder behind a streaming decoder answers while the stream owns its buffered bytes and counters
bool zg_stream_is_finished(const zg::StreamCore* c);
size_t zg_stream_can_collect(const zg::StreamCore* c);
uint64_t zg_stream_blocks_decoded(const zg::StreamCore* c);
uint64_t zg_stream_bytes_read(const zg::StreamCore* c);
bool zg_stream_checksum_from_data(const zg::StreamCore* c, uint32_t* out);
uint32_t zg_stream_calculated_checksum(zg::StreamCore* c);
uint64_t zg_stream_host_bytes(const zg::StreamCore* c);
size_t zg_stream_take(zg::StreamCore* c, uint8_t* dst, size_t n);   // n <= can_collect: bytes that are buffered already
