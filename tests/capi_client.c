/* capi_client.c — the boundary without Python: a plain C program (gcc, links libzgpu.so) that walks the binding of
 * INTEGRATION.md section 2 exactly as the Rust side would, against include/zgpu.h:
 *   zgpu_ctx_create -> zgpu_decoder_create -> zgpu_decoder_init (FrameDecoder::reset) ->
 *   zgpu_decoder_decode_blocks(UptoBytes) + zgpu_decoder_read in a loop (what StreamingDecoder::read drives) ->
 *   checksum_from_data == calculated_checksum,
 * then FrameDecoder::decode_all, decode_all_to_vec, collect_to_writer and the StreamingDecoder mirror over a read callback,
 * and the THIN boundary: this program parses the frame header and every block header itself (what ruzstd's read_frame_header,
 * frame.rs:6-85, and read_block_header, block_decoder.rs:201-247, do on the Rust side) and hands runs of Block_Content to
 * zgpu_frame_begin / zgpu_blocks_submit / zgpu_sync / zgpu_read — the decode_blocks body of INTEGRATION.md section 2 in C.
 * usage: capi_client <file.zst> <expected plaintext file>      exit code 0 = everything agreed
 * Test infrastructure (run by tests/test_gpu_capi_client.py on the GPU box). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/zgpu.h"

static uint8_t* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* b = (uint8_t*)malloc(sz > 0 ? (size_t)sz : 1);
  if (fread(b, 1, (size_t)sz, f) != (size_t)sz) { perror("read"); exit(2); }
  fclose(f);
  *n = (size_t)sz;
  return b;
}
#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, msg); return 1; } } while (0)

typedef struct { const uint8_t* p; size_t n, at; } src_t;
static size_t read_cb(void* user, uint8_t* dst, size_t n) {   /* io::Read::read over a memory source, short reads on purpose */
  src_t* s = (src_t*)user;
  size_t k = s->n - s->at;
  if (k > n) k = n;
  if (k > 70000) k = 70000;
  memcpy(dst, s->p + s->at, k);
  s->at += k;
  return k;
}
typedef struct { uint8_t* p; size_t n, cap; } sink_t;
static size_t write_cb(void* user, const uint8_t* data, size_t n) {
  sink_t* s = (sink_t*)user;
  if (s->n + n > s->cap) n = s->cap - s->n;
  memcpy(s->p + s->n, data, n);
  s->n += n;
  return n;
}


/* ---- the caller's own header parse (RFC 8878 3.1.1.1 / 3.1.1.2; ruzstd frame.rs:6-85, block_decoder.rs:201-247) ---- */
typedef struct { uint64_t window, fcs; uint32_t dict_id; int has_fcs, has_checksum; size_t header_len; } fhdr_t;
static int parse_frame_header(const uint8_t* z, size_t n, fhdr_t* h) {
  if (n < 6 || z[0] != 0x28 || z[1] != 0xB5 || z[2] != 0x2F || z[3] != 0xFD) return -1;
  const uint8_t d = z[4];
  const int fcs_flag = d >> 6, single = (d >> 5) & 1, did_flag = d & 3;
  if (d & 0x08) return -1;                                   /* reserved bit */
  h->has_checksum = (d >> 2) & 1;
  size_t at = 5;
  h->window = 0;
  if (!single) {
    const uint8_t wd = z[at++];
    const uint64_t base = 1ull << (10 + (wd >> 3));
    h->window = base + base / 8 * (wd & 7);
  }
  const int did_len = did_flag == 3 ? 4 : did_flag;
  h->dict_id = 0;
  for (int i = 0; i < did_len; i++) h->dict_id |= (uint32_t)z[at + i] << (8 * i);
  at += did_len;
  const int fcs_len = fcs_flag == 0 ? (single ? 1 : 0) : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
  if (at + fcs_len > n) return -1;
  h->fcs = 0; h->has_fcs = fcs_len != 0;
  for (int i = 0; i < fcs_len; i++) h->fcs |= (uint64_t)z[at + i] << (8 * i);
  if (fcs_len == 2) h->fcs += 256;
  at += fcs_len;
  if (single) h->window = h->fcs;
  h->header_len = at;
  return 0;
}

/* decode one frame through the thin boundary, `run` blocks per submit; returns 0 when the plaintext and the checksum agree */
static int thin_boundary(zgpu_ctx* ctx, const uint8_t* z, size_t zn, const uint8_t* want, size_t pn, uint8_t* out, size_t run) {
  fhdr_t h;
  CHECK(parse_frame_header(z, zn, &h) == 0, "frame header");
  zgpu_frame* fr = NULL;
  int st = zgpu_frame_begin(ctx, h.window, h.has_fcs ? h.fcs : 0, h.dict_id, &fr);
  CHECK(st == ZGPU_OK, zgpu_status_name(st));
  zgpu_block* blk = (zgpu_block*)calloc(run, sizeof(zgpu_block));
  size_t at = h.header_len, got = 0, nblocks = 0;
  int last = 0;
  while (!last) {
    /* read_block_header x run: 3 bytes each, Block_Content behind it */
    const size_t run0 = at;
    size_t k = 0;
    while (k < run && !last) {
      CHECK(at + 3 <= zn, "block header beyond the input");
      const uint32_t bh = z[at] | (uint32_t)z[at + 1] << 8 | (uint32_t)z[at + 2] << 16;
      const uint32_t type = (bh >> 1) & 3, size = bh >> 3;
      CHECK(type != 3, "reserved block type");
      last = bh & 1;
      blk[k].src_off = at + 3 - run0;
      blk[k].src_len = type == 1 ? 1 : size;
      blk[k].raw_rle_size = type == 2 ? 0 : size;
      blk[k].type = (uint8_t)type; blk[k].last = (uint8_t)last;
      at += 3 + blk[k].src_len;
      CHECK(at <= zn, "block content beyond the input");
      k++;
    }
    st = zgpu_blocks_submit(fr, z + run0, at - run0, blk, k);
    CHECK(st == ZGPU_OK, zgpu_status_name(st));
    size_t bad = 0; int32_t bst = 0;
    st = zgpu_sync(fr, &bad, &bst);
    CHECK(st == ZGPU_OK && bad == (size_t)-1 && bst == 0, "zgpu_sync reports a failed block");
    nblocks += k;
    size_t r = 0;
    do {   /* what can_collect allows: everything but the last window while the frame is open */
      CHECK(zgpu_read(fr, out + got, pn + 64 - got, last, &r) == ZGPU_OK, "zgpu_read");
      got += r;
    } while (r);
  }
  CHECK(zgpu_frame_blocks_decoded(fr) == nblocks, "blocks_decoded");
  CHECK(got == pn && memcmp(out, want, pn) == 0, "thin boundary: plaintext differs");
  if (h.has_checksum) {
    CHECK(at + 4 <= zn, "checksum missing");
    const uint32_t cs = z[at] | (uint32_t)z[at + 1] << 8 | (uint32_t)z[at + 2] << 16 | (uint32_t)z[at + 3] << 24;
    CHECK(cs == zgpu_frame_checksum(fr), "thin boundary: content checksum");
  }
  zgpu_frame_end(fr);
  free(blk);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s file.zst plaintext\n", argv[0]); return 2; }
  size_t zn, pn;
  uint8_t* z = slurp(argv[1], &zn);
  uint8_t* want = slurp(argv[2], &pn);
  uint8_t* out = (uint8_t*)malloc(pn + 64);

  zgpu_ctx* ctx = NULL;
  int st = zgpu_ctx_create(0, &ctx);
  CHECK(st == ZGPU_OK, zgpu_status_name(st));

  /* ---- FrameDecoder: reset, decode_blocks(UptoBytes), read ---- */
  zgpu_decoder* d = NULL;
  CHECK(zgpu_decoder_create(ctx, &d) == ZGPU_OK, "decoder_create");
  size_t used = 0, pos = 0, got = 0;
  uint32_t sm = 0, sl = 0;
  st = zgpu_decoder_init(d, z, zn, &used, &sm, &sl);
  CHECK(st == ZGPU_OK, zgpu_status_name(st));
  pos = used;
  int fin = 0;
  while (!zgpu_decoder_is_finished(d)) {
    size_t c = 0;
    st = zgpu_decoder_decode_blocks(d, z + pos, zn - pos, &c, ZGPU_STRAT_UPTO_BYTES, 300000, &fin);
    CHECK(st == ZGPU_OK, zgpu_status_name(st));
    pos += c;
    size_t r;
    while ((r = zgpu_decoder_read(d, out + got, pn + 64 - got)) > 0) got += r;
  }
  { size_t r; while ((r = zgpu_decoder_read(d, out + got, pn + 64 - got)) > 0) got += r; }
  CHECK(got == pn && memcmp(out, want, pn) == 0, "streamed plaintext differs");
  CHECK(zgpu_decoder_bytes_read_from_source(d) == zn, "bytes_read_from_source");
  uint32_t cs = 0;
  if (zgpu_decoder_checksum_from_data(d, &cs)) CHECK(cs == zgpu_decoder_calculated_checksum(d), "content checksum");

  /* ---- collect_to_writer ---- */
  st = zgpu_decoder_init(d, z, zn, &used, &sm, &sl);
  CHECK(st == ZGPU_OK, "re-init");
  { size_t c = 0; st = zgpu_decoder_decode_blocks(d, z + used, zn - used, &c, ZGPU_STRAT_ALL, 0, &fin); }
  CHECK(st == ZGPU_OK && fin, "decode_blocks(All)");
  sink_t sink = {out, 0, pn + 64};
  size_t wrote = 0;
  CHECK(zgpu_decoder_collect_to_writer(d, write_cb, &sink, &wrote) == ZGPU_OK, "collect_to_writer");
  CHECK(wrote == pn && sink.n == pn && memcmp(out, want, pn) == 0, "collect_to_writer plaintext differs");
  zgpu_decoder_destroy(d);

  /* ---- FrameDecoder::decode_all and decode_all_to_vec ---- */
  size_t w = 0;
  memset(out, 0, pn);
  st = zgpu_decode_all(ctx, z, zn, out, pn, &w);
  CHECK(st == ZGPU_OK && w == pn && memcmp(out, want, pn) == 0, "decode_all");
  if (pn > 1) CHECK(zgpu_decode_all(ctx, z, zn, out, pn - 1, &w) == ZGPU_E_TARGET_TOO_SMALL, "TargetTooSmall");
  uint8_t* vec = NULL;
  st = zgpu_decode_all_alloc(ctx, z, zn, &vec, &w);
  CHECK(st == ZGPU_OK && w == pn && memcmp(vec, want, pn) == 0, "decode_all_to_vec");
  zgpu_free(vec);

  /* ---- StreamingDecoder over a read callback ---- */
  src_t src = {z, zn, 0};
  zgpu_streaming* sd = NULL;
  st = zgpu_streaming_create(ctx, read_cb, &src, &sd);
  CHECK(st == ZGPU_OK, zgpu_status_name(st));
  got = 0;
  for (;;) {
    size_t n = 0;
    st = zgpu_streaming_read(sd, out + got, (pn + 64 - got) < 1000003 ? (pn + 64 - got) : 1000003, &n);
    CHECK(st == ZGPU_OK, zgpu_status_name(st));
    if (n == 0) break;
    got += n;
  }
  CHECK(got == pn && memcmp(out, want, pn) == 0, "StreamingDecoder plaintext differs");
  CHECK(src.at == zn, "StreamingDecoder must consume exactly the frame");
  zgpu_streaming_destroy(sd);

  /* ---- the thin boundary: own header parse, block runs of 1, 7 and 4096 blocks per submit ---- */
  if (thin_boundary(ctx, z, zn, want, pn, out, 7) || thin_boundary(ctx, z, zn, want, pn, out, 4096) ||
      (zn < (1u << 20) && thin_boundary(ctx, z, zn, want, pn, out, 1))) return 1;

  /* ---- the work queue on the GPUs of this box ---- */
  zgpu_pool* pool = NULL;
  CHECK(zgpu_pool_create(0, &pool) == ZGPU_OK && zgpu_pool_num_gpus(pool) >= 1, "pool_create");
  memset(out, 0, pn);
  st = zgpu_pool_decode_all(pool, z, zn, out, pn, &w);
  CHECK(st == ZGPU_OK && w == pn && memcmp(out, want, pn) == 0, "pool_decode_all");
  zgpu_pool_destroy(pool);

  zgpu_ctx_destroy(ctx);
  printf("capi_client ok: %zu -> %zu bytes, every surface agreed\n", zn, pn);
  free(z); free(want); free(out);
  return 0;
}
