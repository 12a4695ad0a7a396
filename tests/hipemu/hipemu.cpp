// hipemu scheduler: runs the work-items of one workgroup as fibers on the calling thread.
// TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <vector>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {
State g;
static void* sched_sp = nullptr;
static const std::function<void()>* body_fn = nullptr;
static constexpr size_t kStack = 256 * 1024;
static std::vector<Fiber> fibers;
static std::vector<Wave> waves_store;
static std::vector<char*> stacks;

static inline void yield_to_sched() { hipemu_switch(&g.cur->sp, sched_sp); }

static void release(Barrier& b) { b.count = 0; b.gen++; }

static void fiber_exit() {
  Fiber* f = g.cur;
  f->done = true;
  g.alive--;
  Wave& w = g.waves[f->wave];
  w.alive--;
  // a barrier that was only waiting for this work-item is now complete
  if (g.alive > 0 && g.blockbar.count == g.alive) release(g.blockbar);
  if (w.alive > 0 && w.bar.count == w.alive) { w.parity++; release(w.bar); }
  yield_to_sched();
  fprintf(stderr, "hipemu: resumed a finished fiber\n");
  abort();
}

static void trampoline() {
  (*body_fn)();
  fiber_exit();
}

void block_sync() {
  Barrier& b = g.blockbar;
  unsigned my = b.gen;
  if (++b.count == g.alive) { release(b); return; }
  g.cur->wait = &b;
  g.cur->wait_gen = my;
  yield_to_sched();
}

void wave_sync() {
  Wave& w = wave();
  Barrier& b = w.bar;
  unsigned my = b.gen;
  if (++b.count == w.alive) { w.parity++; release(b); return; }
  g.cur->wait = &b;
  g.cur->wait_gen = my;
  yield_to_sched();
}

static char* get_stack(size_t i) {
  while (stacks.size() <= i) {
    void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    stacks.push_back(static_cast<char*>(p));
  }
  return stacks[i];
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const int nthreads = int(block.x * block.y * block.z);
  const int nwaves = (nthreads + 63) / 64;
  if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
  fibers.assign(nthreads, Fiber());
  waves_store.assign(nwaves, Wave());
  std::vector<char> dyn(shmem + 64);
  g.bdim = block;
  g.gdim = grid;
  g.waves = waves_store.data();
  g.dynsh = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(dyn.data()) + 63) & ~uintptr_t(63));
  body_fn = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g.bid = uint3{bx, by, bz};
        g.blockbar = Barrier();
        g.alive = nthreads;
        for (int w = 0; w < nwaves; ++w) {
          waves_store[w].bar = Barrier();
          waves_store[w].parity = 0;
          waves_store[w].alive = (w == nwaves - 1) ? nthreads - 64 * w : 64;
        }
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = fibers[t];
          f = Fiber();
          f.stack = get_stack(t);
          f.lin = t;
          f.wave = t / 64;
          f.lane = t % 64;
          f.tid = uint3{unsigned(t % block.x), unsigned((t / block.x) % block.y), unsigned(t / (block.x * block.y))};
          uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
          void** sp = reinterpret_cast<void**>(top);
          *--sp = nullptr;                                   // fake return address (rsp%16==8 at entry)
          *--sp = reinterpret_cast<void*>(&trampoline);      // ret target
          for (int r = 0; r < 6; ++r) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
          f.sp = sp;
        }
        int remaining = nthreads;
        while (remaining > 0) {
          bool progress = false;
          for (int t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[t];
            if (f.done) continue;
            if (f.wait) {
              if (f.wait->gen == f.wait_gen) continue;       // still blocked
              f.wait = nullptr;
            }
            g.cur = &f;
            hipemu_switch(&sched_sp, f.sp);
            progress = true;
            if (f.done) remaining--;
          }
          if (!progress) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d work-items blocked on a barrier that "
                            "cannot complete (divergent __syncthreads / partial-wave collective)\n", bx, by, bz, remaining);
            abort();
          }
        }
      }
  g.cur = nullptr;
}
}  // namespace hipemu
