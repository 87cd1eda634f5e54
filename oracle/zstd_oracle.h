/*
 * zstd_oracle.h — CPU oracle for the zgpu block-decode engine.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C restatement of the decode path of the
 * reference implementation (KillingSpark/zstd-rs, crate `ruzstd` 0.9.1, /root/reference).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (libzgpu.so) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED — checked against the reference's own golden fixtures
 * (101 decodecorpus pairs, 207 dictionary pairs, window fixtures, KATs); see
 * tests/test_oracle_*.py and tests/golden/make_golden.py.
 *
 * Every function in zstd_oracle.c cites the reference file:line it follows
 * (paths relative to /root/reference/ruzstd/src).
 */
#ifndef ZSTD_ORACLE_H
#define ZSTD_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Status codes: leaves of the reference's error enums (decoding/errors.rs). 0 = ok. */
enum zor_status {
  ZOR_OK = 0,
  /* frame layer (errors.rs: ReadFrameHeaderError / FrameHeaderError / FrameDecoderError) */
  ZOR_SKIP_FRAME = 1,            /* ReadFrameHeaderError::SkipFrame (returned as error by design, frame.rs:15-23) */
  ZOR_BAD_MAGIC = 2,
  ZOR_HEADER_READ = 3,           /* any *ReadError while reading the frame header */
  ZOR_WINDOW_TOO_BIG_SPEC = 4,   /* FrameHeaderError::WindowTooBig */
  ZOR_WINDOW_TOO_SMALL = 5,
  ZOR_WINDOW_SIZE_TOO_BIG = 6,   /* FrameDecoderError::WindowSizeTooBig (over the configured limit) */
  ZOR_DICT_NOT_PROVIDED = 7,
  ZOR_NOT_INITIALIZED = 8,
  ZOR_FAILED_READ_BLOCK_HEADER = 9,
  ZOR_FAILED_READ_BLOCK_BODY = 10,  /* short read of a block body */
  ZOR_FAILED_READ_CHECKSUM = 11,
  ZOR_TARGET_TOO_SMALL = 12,
  ZOR_FAILED_SKIP_FRAME = 13,
  /* block layer */
  ZOR_RESERVED_BLOCK = 20,
  ZOR_BLOCK_SIZE_TOO_LARGE = 21,
  ZOR_MALFORMED_SECTION_HEADER = 22,
  ZOR_LITERALS_HEADER = 23,      /* LiteralsSectionParseError */
  ZOR_SEQUENCES_HEADER = 24,     /* SequencesHeaderParseError */
  /* literals (DecompressLiteralsError / HuffmanTableError) */
  ZOR_LIT_UNINIT_HUF = 30,
  ZOR_LIT_MISSING_JUMP = 31,
  ZOR_LIT_MISSING_BYTES = 32,
  ZOR_LIT_EXTRA_PADDING = 33,
  ZOR_LIT_BITSTREAM_MISMATCH = 34,
  ZOR_LIT_COUNT_MISMATCH = 35,
  ZOR_HUF_TABLE = 36,            /* any HuffmanTableError leaf */
  /* sequences (DecodeSequenceError / FSETableError / FSEDecoderError) */
  ZOR_FSE_TABLE = 40,            /* any FSETableError leaf */
  ZOR_FSE_UNINIT = 41,           /* FSEDecoderError::TableIsUninitialized */
  ZOR_SEQ_MISSING_MODE = 42,
  ZOR_SEQ_RLE_BYTE = 43,         /* MissingByteForRle*Table (also out-of-range RLE symbol) */
  ZOR_SEQ_EXTRA_PADDING = 44,
  ZOR_SEQ_UNSUPPORTED_OFFSET = 45,
  ZOR_SEQ_NOT_ENOUGH_BYTES = 46, /* NotEnoughBytesForNumSequences */
  ZOR_SEQ_EXTRA_BITS = 47,
  /* execution (ExecuteSequencesError / DecodeBufferError) */
  ZOR_EXE_NOT_ENOUGH_LITERALS = 50,
  ZOR_EXE_ZERO_OFFSET = 51,
  ZOR_EXE_OFFSET_TOO_BIG = 52,
  ZOR_EXE_DICT_TOO_SMALL = 53,
  /* dictionary */
  ZOR_DICT_DECODE = 60,
  /* places where the reference would panic/assert (kept distinct so tests can see them) */
  ZOR_REF_PANIC = 90,
  ZOR_NOMEM = 91
};

typedef struct zor_decoder zor_decoder; /* mirrors FrameDecoder (frame_decoder.rs:80-84) */

/* one decoded sequence, as the reference's Sequence (blocks/sequence_section.rs:21-37) plus the
 * resolved offset computed by do_offset_history (sequence_execution.rs:59-118) */
typedef struct { uint32_t ll, ml, of, actual_of; } zor_sequence;

/* FSE entry as fse_decoder.rs:312-320 */
typedef struct { uint32_t base_line; uint8_t num_bits; uint8_t symbol; uint8_t pad[2]; } zor_fse_entry;
/* Huffman entry as huff0_decoder.rs:389-394 */
typedef struct { uint8_t symbol; uint8_t num_bits; } zor_huf_entry;

enum { ZOR_STRAT_ALL = 0, ZOR_STRAT_UPTO_BLOCKS = 1, ZOR_STRAT_UPTO_BYTES = 2 };

zor_decoder* zor_new(void);                                   /* FrameDecoder::new  frame_decoder.rs:158 */
void zor_free(zor_decoder*);
void zor_set_max_window_size(zor_decoder*, uint64_t);         /* frame_decoder.rs:175 */
int  zor_add_dict(zor_decoder*, const uint8_t* raw, size_t len, uint32_t* id_out); /* Dictionary::decode_dict + add_dict */
int  zor_force_dict(zor_decoder*, uint32_t id);               /* frame_decoder.rs:229 */

/* init/reset: parse a frame header from src. *consumed = header bytes. On ZOR_SKIP_FRAME,
 * *skip_magic / *skip_len are filled and *consumed = 8 (frame_decoder.rs:190-221, frame.rs:6-85). */
int  zor_init(zor_decoder*, const uint8_t* src, size_t len, size_t* consumed,
              uint32_t* skip_magic, uint32_t* skip_len);
/* decode_blocks (frame_decoder.rs:309-377). *consumed = bytes taken from src. */
int  zor_decode_blocks(zor_decoder*, const uint8_t* src, size_t len, size_t* consumed,
                       int strat, size_t n, int* frame_finished);
size_t zor_can_collect(const zor_decoder*);                   /* frame_decoder.rs:410-424 */
size_t zor_collect(zor_decoder*, uint8_t* dst, size_t cap);   /* collect(): drain all / drain to window; returns bytes */
size_t zor_read(zor_decoder*, uint8_t* dst, size_t cap);      /* impl Read  frame_decoder.rs:615-627 */
/* test accessor without a counterpart in the reference: a copy of what the decode buffer holds (nothing is drained or hashed) */
size_t zor_held(const zor_decoder*, uint8_t* dst, size_t cap);
/* decode_from_to (frame_decoder.rs:439-529): *read_out = bytes taken from src, *written_out = bytes drained to dst */
int  zor_decode_from_to(zor_decoder*, const uint8_t* src, size_t len, uint8_t* dst, size_t cap, size_t* read_out, size_t* written_out);
int  zor_decode_all(zor_decoder*, const uint8_t* in, size_t inlen, uint8_t* out, size_t outcap,
                    size_t* written);                         /* frame_decoder.rs:541-577 */

int      zor_is_finished(const zor_decoder*);                 /* :284-294 */
uint64_t zor_blocks_decoded(const zor_decoder*);              /* :297 */
uint64_t zor_bytes_read_from_source(const zor_decoder*);      /* :273 */
uint64_t zor_content_size(const zor_decoder*);                /* :246 */
uint64_t zor_window_size(const zor_decoder*);
int      zor_checksum_from_data(const zor_decoder*, uint32_t* out);  /* :254  returns 1 if present */
uint32_t zor_calculated_checksum(const zor_decoder*);         /* :263-270 (XXH64 seed 0, low 32 bits) */
uint32_t zor_dict_id(const zor_decoder*);                     /* 0 = none */

/* ---- intermediates of the most recently decoded block (for kernel-level parity tests) ---- */
int      zor_last_block_type(const zor_decoder*);             /* 0 raw, 1 rle, 2 compressed */
const uint8_t* zor_last_literals(const zor_decoder*, size_t* len);
const zor_sequence* zor_last_sequences(const zor_decoder*, size_t* n);
void     zor_offset_hist(const zor_decoder*, uint32_t out[3]);
/* which: 0 = LL, 1 = OF, 2 = ML. Returns table size (0 if uninitialised); *rle = symbol or -1 */
size_t   zor_fse_table(const zor_decoder*, int which, const zor_fse_entry** entries, int* acc_log, int* rle);
size_t   zor_huf_table(const zor_decoder*, const zor_huf_entry** entries, int* max_bits);

/* ---- stand-alone pieces, for KATs ---- */
/* FSETable::build_from_probabilities (fse_decoder.rs:126-139); out must hold 1<<acc_log entries */
int zor_fse_build_from_probs(int acc_log, const int32_t* probs, size_t nprobs, int max_symbol, zor_fse_entry* out);
/* FSETable::build_decoder (fse_decoder.rs:116-124): returns status; *bytes_read, *acc_log filled; out holds ≤512 */
int zor_fse_build_decoder(const uint8_t* src, size_t len, int max_log, int max_symbol,
                          zor_fse_entry* out, int* acc_log, size_t* bytes_read);
/* HuffmanTable::build_decoder (huff0_decoder.rs:117-124); out holds ≤2048 */
int zor_huf_build_decoder(const uint8_t* src, size_t len, zor_huf_entry* out, int* max_bits, uint32_t* bytes_read);
/* do_offset_history (sequence_execution.rs:59-118) */
uint32_t zor_do_offset_history(uint32_t offset_value, uint32_t lit_len, uint32_t hist[3]);
/* BitReaderReversed::get_bits sequence (bit_reader_reverse.rs:92-100): reads widths[i] bits each, writes values;
 * returns bits_remaining() afterwards */
int64_t zor_revbits_read(const uint8_t* src, size_t len, const uint8_t* widths, size_t n, uint64_t* values);
/* BitReader::get_bits sequence (bit_reader.rs:28-91); returns 0 or -1 on NotEnoughRemainingBits */
int zor_fwdbits_read(const uint8_t* src, size_t len, const uint8_t* widths, size_t n, uint64_t* values);
/* XXH64 (twox-hash 2.x XxHash64, seed given) */
uint64_t zor_xxh64(const uint8_t* p, size_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
