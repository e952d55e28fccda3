// hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.h).  Fiber scheduler for the SIMT emulator.
#include "hip_emu.h"
#include <sys/mman.h>

namespace emu {

thread_local Block* g_blk = nullptr;

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
  swapcontext(&f->ctx, &b->sched);
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
static thread_local StackPool g_pool;

void run_block(Block& b) {
  g_blk = &b;
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
    f.stack = g_pool.get(i);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  int remaining = n;
  while (remaining > 0) {
    b.progress = false;
    remaining = 0;
    for (int i = 0; i < n; ++i) {
      Fiber& f = b.fibers[i];
      if (f.done) continue;
      b.cur = &f;
      swapcontext(&b.sched, &f.ctx);
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

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  size_t total = (size_t)grid.x * grid.y * grid.z;
  if (total == 0) return;
  unsigned nthr = std::thread::hardware_concurrency();
  if (const char* e = getenv("VDK_EMU_THREADS")) nthr = (unsigned)atoi(e);
  if (nthr < 1) nthr = 1;
  if (nthr > total) nthr = (unsigned)total;
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    Block b;
    b.bdim = block; b.gdim = grid; b.body = &body;
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= total) break;
      b.bid.x = (unsigned)(i % grid.x); b.bid.y = (unsigned)((i / grid.x) % grid.y); b.bid.z = (unsigned)(i / ((size_t)grid.x * grid.y));
      run_block(b);
    }
  };
  if (nthr == 1) { worker(); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthr; ++t) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

}  // namespace emu
