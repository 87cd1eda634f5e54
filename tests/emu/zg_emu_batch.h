// zg_emu_batch.h — TEST-ONLY: the state of one submit in the CPU harness (zg_emu.cpp fills it stage by stage, in the
// order of the HIP pipeline; zg_emu_flat.cpp runs the flatten kernel's source on it).
#pragma once
#include <stdint.h>
#include <vector>
#include "zg_emu_serial.h"
#include "../../zstd-rs_amd/csrc/zg_host_parse.h"

struct EmuBatch {
  zg::BatchBuilder bb;
  std::vector<uint8_t> src_store;   // 64 bytes of padding in front and behind, as the engine allocates it
  uint8_t* src = nullptr;
  int use_fast = 1;
  std::vector<ZgBlockAux> aux;
  std::vector<uint8_t> slot_log;
  std::vector<uint32_t> fse;
  std::vector<uint16_t> huf;
  std::vector<uint8_t> hufmax;
  std::vector<uint32_t> status;
  std::vector<uint32_t> fse_status;   // verdicts of the FSE table descriptions: they come behind the literals' (zg_k_merge)
  std::vector<uint8_t> lit;
  std::vector<EmuSeq> seq;
  std::vector<ZgBlockSeqOut> seqout;
  std::vector<ZgBlockPos> pos;
  std::vector<ZgFrameOut> fout;
  std::vector<uint8_t> dst;
  int parse_status = 0;
};
