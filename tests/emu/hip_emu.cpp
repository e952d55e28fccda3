// hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.h).  Fiber scheduler for the SIMT emulator.
#include "hip_emu.h"
#include <sys/mman.h>
#include <mutex>

// void emu_switch(Ctx* from, Ctx* to): save the System V callee-saved registers on the current stack, swap stack pointers, restore, return
__asm__(
    ".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq (%rsi), %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size emu_switch,.-emu_switch\n");

namespace emu {

thread_local Block* g_blk = nullptr;

static void fiber_trampoline() {
  fiber_entry();          // never returns: a finished fiber switches back to the scheduler for good
  abort();
}

void fiber_entry() {
  Block* b = g_blk;
  Fiber* f = b->cur;
  (*b->body)();
  f->done = true;
  b->progress = true;
  // leaving threads no longer take part in barriers
  b->alive--;
  Wave& w = b->waves[f->lin >> 6];
  w.alive--;
  if (b->alive > 0 && b->bar.arrived >= b->alive && b->bar.arrived > 0) { b->bar.arrived = 0; b->bar.gen++; }
  if (w.alive > 0 && w.bar.arrived >= w.alive && w.bar.arrived > 0) { w.bar.arrived = 0; w.bar.gen++; }
  emu_switch(&f->ctx, &b->sched);
}

struct StackPool {
  std::vector<char*> stacks;
  char* get(size_t i) {
    while (stacks.size() <= i) {
      void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (p == MAP_FAILED) { perror("mmap"); abort(); }
      stacks.push_back((char*)p);
    }
    return stacks[i];
  }
};
// One pool per worker slot, kept for the life of the process (launches are serialised by g_launch_mu): the stacks are mapped once, not once per
// launch and thread.
static StackPool g_pools[256];
static thread_local int g_slot = 0;
static std::mutex g_launch_mu;

static const size_t kDynLds = 160 * 1024;
static thread_local unsigned char* g_dyn = nullptr;
unsigned char* dyn_lds() {
  if (!g_dyn) { if (posix_memalign((void**)&g_dyn, 256, kDynLds)) abort(); }
  return g_dyn;
}

void run_block(Block& b) {
  g_blk = &b;
  if (b.dyn_lds_bytes) {
    if (b.dyn_lds_bytes > kDynLds) { fprintf(stderr, "[hip_emu] dynamic LDS request %zu > 160 KB\n", b.dyn_lds_bytes); abort(); }
    unsigned short* p = (unsigned short*)dyn_lds();
    for (size_t i = 0; i < kDynLds / 2; ++i) p[i] = 0x7fc0;   // bf16 NaN (0x7fc07fc0 is an f32 NaN too)
  }
  int n = (int)(b.bdim.x * b.bdim.y * b.bdim.z);
  b.fibers.assign(n, Fiber());
  b.waves.assign((n + 63) / 64, Wave());
  b.alive = n;
  b.bar = Barrier();
  for (int i = 0; i < n; ++i) {
    Fiber& f = b.fibers[i];
    f.blk = &b; f.lin = i; f.done = false;
    f.tid.x = i % b.bdim.x; f.tid.y = (i / b.bdim.x) % b.bdim.y; f.tid.z = i / (b.bdim.x * b.bdim.y);
    b.waves[i >> 6].alive++;
    f.stack = g_pools[g_slot].get(i);
    // initial frame: six zeroed callee-saved registers + the trampoline as return address; after the first `ret` rsp == top, and top is 8 mod 16
    // as the ABI expects at a function's first instruction
    uintptr_t top = ((uintptr_t)(f.stack + kStack) & ~(uintptr_t)15) - 8;
    void** sp = (void**)top;
    *--sp = (void*)fiber_trampoline;
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    f.ctx.sp = sp;
  }
  int remaining = n;
  while (remaining > 0) {
    b.progress = false;
    remaining = 0;
    for (int i = 0; i < n; ++i) {
      Fiber& f = b.fibers[i];
      if (f.done) continue;
      b.cur = &f;
      emu_switch(&b.sched, &f.ctx);
      if (!f.done) remaining++;
    }
    if (remaining > 0 && !b.progress) {
      fprintf(stderr, "[hip_emu] deadlock: %d fibers blocked (divergent barrier/collective?) block=(%u,%u,%u)\n",
              remaining, b.bid.x, b.bid.y, b.bid.z);
      abort();
    }
  }
  g_blk = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t dyn_lds_bytes) {
  size_t total = (size_t)grid.x * grid.y * grid.z;
  if (total == 0) return;
  unsigned nthr = std::thread::hardware_concurrency();
  if (const char* e = getenv("VDK_EMU_THREADS")) nthr = (unsigned)atoi(e);
  if (nthr < 1) nthr = 1;
  if (nthr > total) nthr = (unsigned)total;
  if (nthr > 256) nthr = 256;
  std::lock_guard<std::mutex> lk(g_launch_mu);
  std::atomic<size_t> next{0};
  auto worker = [&](int slot) {
    g_slot = slot;
    Block b;
    b.bdim = block; b.gdim = grid; b.body = &body; b.dyn_lds_bytes = dyn_lds_bytes;
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= total) break;
      b.bid.x = (unsigned)(i % grid.x); b.bid.y = (unsigned)((i / grid.x) % grid.y); b.bid.z = (unsigned)(i / ((size_t)grid.x * grid.y));
      run_block(b);
    }
  };
  if (nthr == 1) { worker(0); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthr; ++t) th.emplace_back(worker, (int)t);
  for (auto& t : th) t.join();
}

}  // namespace emu
