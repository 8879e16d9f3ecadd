// tests/native/wave_emu.h -- TEST INFRASTRUCTURE ONLY.  Never part of the product.
//
// Runs wave-cooperative device code (svdss_amd/csrc/poa_quad_core.h) on the CPU: the 64 lanes of a wavefront are 64
// cooperative fibres that run freely between cross-lane primitives and meet at every one of them (shift, scan,
// ballot, permute, the two "sync" markers).  A primitive deposits the lane's operand, lets the other lanes arrive at
// the same primitive -- the call sites are compared, so code whose lanes would meet different instructions is
// reported instead of silently "working" -- and computes the lane's result from the 64 deposits.  That makes the
// emulator STRICTER than the hardware in one way (every cross-lane operation has to sit in wave-uniform control flow)
// and exactly as strict in the way that matters: lanes communicate only through the primitives and through memory
// separated by a sync marker, so a race the hardware resolves by lock-step execution shows up here as a mismatch
// with the oracle.  LDS is one buffer per emulated block; global memory is host memory; atomics are plain
// operations (fibres are not pre-empted).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace wemu {

constexpr int W = 64;
struct Fiber { void* sp = nullptr; std::vector<unsigned char> stack; bool done = false; };

extern "C" void wemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl wemu_switch
.type wemu_switch,@function
wemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size wemu_switch,.-wemu_switch
)");

struct Wave {
  Fiber f[W];
  void* main_sp = nullptr;
  int cur = -1;          // running lane
  int block = 0;
  // deposits of the two most recent primitives (a lane that ran ahead writes the other buffer)
  uint64_t dep[2][W];
  int site[2][W];
  unsigned seq[W];
  void (*body)(void*) = nullptr;
  void* arg = nullptr;
  unsigned char* lds = nullptr;
  long n_prims = 0;
};

inline Wave*& cur_wave() { static Wave* w = nullptr; return w; }
inline int lane_id() { return cur_wave()->cur; }
inline int block_id() { return cur_wave()->block; }
inline unsigned char* lds_base() { return cur_wave()->lds; }

// lane -> next lane that is not done; the last lane of a round hands over to lane 0 (round-robin: every lane has
// passed primitive k before any lane passes primitive k + 2)
inline void yield_lane() {
  Wave* w = cur_wave();
  const int me = w->cur;
  int nx = me;
  for (int t = 0; t < W; ++t) { nx = (nx + 1) % W; if (!w->f[nx].done) break; }
  if (nx == me && !w->f[me].done) return;
  if (w->f[nx].done) {           // everybody is done: back to the launcher
    w->cur = -1;
    wemu_switch(&w->f[me].sp, w->main_sp);
    return;
  }
  w->cur = nx;
  wemu_switch(&w->f[me].sp, w->f[nx].sp);
}

inline void fiber_entry() {
  Wave* w = cur_wave();
  w->body(w->arg);
  w->f[w->cur].done = true;
  yield_lane();
  abort();   // never resumed
}

// deposit v at call site `s`, wait for the other lanes, return the buffer of the 64 deposits
inline const uint64_t* meet(uint64_t v, int s) {
  Wave* w = cur_wave();
  const int me = w->cur;
  const unsigned k = w->seq[me]++;
  w->dep[k & 1][me] = v;
  w->site[k & 1][me] = s;
  if (me == 0) ++w->n_prims;
  yield_lane();
  // (a lane can be at most one primitive ahead: it has written the other buffer; a lane that left the kernel before this
  // primitive never deposited)
  for (int i = 0; i < W; ++i) {
    if (w->seq[i] <= k || w->site[k & 1][i] != s) {
      fprintf(stderr, "wave_emu: lanes diverge at a cross-lane primitive: lane %d at site %d, lane %d %s (site %d)\n", me, s, i,
              w->seq[i] <= k ? "never arrived" : "is elsewhere", w->site[k & 1][i]);
      abort();
    }
  }
  return w->dep[k & 1];
}

// runs body(arg) on 64 lanes of block `block` with `lds_bytes` of zero-initialised LDS... (the hardware leaves LDS
// undefined: the emulator fills it with a pattern so that code relying on zeros fails)
inline void run_block(int block, size_t lds_bytes, void (*body)(void*), void* arg, size_t stack_bytes = 1 << 20) {
  Wave w;
  std::vector<unsigned char> lds(lds_bytes + 64, 0xA5);
  w.lds = lds.data();
  w.block = block;
  w.body = body;
  w.arg = arg;
  memset(w.seq, 0, sizeof w.seq);
  memset(w.site, 0xff, sizeof w.site);
  for (int i = 0; i < W; ++i) {
    w.f[i].stack.resize(stack_bytes);
    uintptr_t top = (uintptr_t)(w.f[i].stack.data() + stack_bytes);
    top &= ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                       // keeps rsp = 8 mod 16 at the entry, as after a call
    *--sp = (void*)&fiber_entry;           // return address popped by wemu_switch's ret
    for (int k = 0; k < 6; ++k) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    w.f[i].sp = sp;
  }
  Wave* prev = cur_wave();
  cur_wave() = &w;
  w.cur = 0;
  wemu_switch(&w.main_sp, w.f[0].sp);
  cur_wave() = prev;
}

}  // namespace wemu
