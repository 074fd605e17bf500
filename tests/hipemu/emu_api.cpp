// Extra entry points of libtamd_emu.so (TEST INFRASTRUCTURE): LDS bank-conflict statistics.
#include "hipemu.h"

extern "C" {
// out[0..7] = read16 instr, read16 cycles, tr instr, tr cycles, write8 instr, write8 cycles, mfma instr, 0
void hipemu_get_stats(unsigned long long* out) {
  auto& s = hipemu::g_stats;
  out[0] = s.lds_read16_instr;
  out[1] = s.lds_read16_cycles;
  out[2] = s.lds_tr_instr;
  out[3] = s.lds_tr_cycles;
  out[4] = s.lds_write8_instr;
  out[5] = s.lds_write8_cycles;
  out[6] = s.mfma_instr;
  out[7] = 0;
}
void hipemu_reset_stats() {
  auto& s = hipemu::g_stats;
  s.lds_read16_instr = 0;
  s.lds_read16_cycles = 0;
  s.lds_tr_instr = 0;
  s.lds_tr_cycles = 0;
  s.lds_write8_instr = 0;
  s.lds_write8_cycles = 0;
  s.mfma_instr = 0;
}
}
