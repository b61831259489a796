// tests/emu/emu.cpp — TEST INFRASTRUCTURE ONLY (see tests/emu/hip/hip_runtime.h).
//
// Executes a "kernel launch" on the host: the blocks of the grid one after the other, the threads of a block as fibers
// on one host thread.  A block that waits for OTHER blocks (the grid barrier of the persistent round-tail kernel: its
// polling lane calls emu::grid_yield() where the GPU code sleeps) is parked with its fibers, stacks and dynamic LDS intact,
// the remaining blocks are started, and the parked ones are resumed in turn until every block has finished - so a kernel
// whose blocks synchronise through global memory runs here with as many co-resident blocks as its grid has.
// A fiber runs until it finishes or reaches a rendezvous:
//   * a cross-lane operation of its wave (shuffle / DPP shift / ballot / readfirstlane) over the lanes
//     [base, base + width) - released when every LIVE lane of that group waits in an operation of the same width, and
//     the scheduler computes each lane's result at that moment (lanes that have left the kernel count as inactive:
//     their values read as invalid, as an EXEC-masked lane's would);
//   * __syncthreads() - released when every live thread of the block waits in it.
// Narrower groups are released before wider ones, which is the order a real wave executes divergent code in (the
// 16-lane shuffles of one lane group inside a branch other groups do not take, then the wave-wide operation behind the
// branch).  A state in which nobody can run and nobody can be released is reported as a deadlock with the waiting lanes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <csignal>
#include <execinfo.h>
#include <dlfcn.h>
#include <ucontext.h>
#include <cstring>
#include <unistd.h>
#include <mutex>
#include <sys/mman.h>
#include <vector>

namespace emu {

Cur cur;

// a crash inside an emulated kernel: print the native frames before dying (fibers run on their own stacks, so the handler
// gets an alternate one)
static void put_hex(const char *label, unsigned long v) {
  char buf[64];
  int n = 0;
  while (label[n]) { buf[n] = label[n]; n++; }
  for (int k = 60; k >= 0; k -= 4) buf[n++] = "0123456789abcdef"[(v >> k) & 15];
  buf[n++] = '\n';
  (void)!write(2, buf, n);
}
static void on_segv(int sig, siginfo_t *si, void *uc) {
  const char msg[] = "[emu] fatal signal inside the emulated library\n";
  (void)!write(2, msg, sizeof msg - 1);
  // first what needs no unwinding (a fault inside a kernel runs on a fiber stack the unwinder may not get through):
  // the address touched, the program counter, and where this library is loaded (pc - base goes to addr2line)
  put_hex("[emu]   address 0x", (unsigned long)si->si_addr);
  const unsigned long pc = (unsigned long)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
  put_hex("[emu]   pc      0x", pc);
  Dl_info info;
  if (dladdr((void *)pc, &info) && info.dli_fbase) {
    put_hex("[emu]   pc - library base 0x", pc - (unsigned long)info.dli_fbase);
    if (info.dli_sname) { (void)!write(2, "[emu]   in ", 11); (void)!write(2, info.dli_sname, strlen(info.dli_sname)); (void)!write(2, "\n", 1); }
  }
  put_hex("[emu]   block.x 0x", cur.bid.x);
  put_hex("[emu]   thread.x 0x", cur.tid.x);
  alarm(5);                                    // (should the unwinder loop on a fiber stack)
  void *frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
static const int segv_installed = [] {
  static char alt[64 * 1024];
  stack_t ss{};
  ss.ss_sp = alt; ss.ss_size = sizeof alt;
  sigaltstack(&ss, nullptr);
  struct sigaction sa{};
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_ONSTACK | SA_SIGINFO;
  sigaction(SIGSEGV, &sa, nullptr);
  sigaction(SIGBUS, &sa, nullptr);
  return 1;
}();

// guarded allocations (EMU_GUARD=1): [pages ... | buffer, 16-byte aligned end at the page boundary][PROT_NONE page]
static std::mutex guard_mu;
static std::vector<std::pair<void *, std::pair<void *, size_t>>> guard_live;   // user pointer -> (mapping, length)
void *guard_alloc(size_t n) {
  const size_t page = 4096, body = (n + 15) & ~(size_t)15, len = ((body + page - 1) & ~(page - 1)) + page;
  char *m = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == (char *)MAP_FAILED) return nullptr;
  mprotect(m + len - page, page, PROT_NONE);
  void *user = m + len - page - body;
  std::lock_guard<std::mutex> g(guard_mu);
  guard_live.push_back({user, {m, len}});
  return user;
}
bool guard_free(void *p) {
  std::lock_guard<std::mutex> g(guard_mu);
  for (size_t i = 0; i < guard_live.size(); i++)
    if (guard_live[i].first == p) {
      munmap(guard_live[i].second.first, guard_live[i].second.second);
      guard_live[i] = guard_live.back();
      guard_live.pop_back();
      return true;
    }
  return false;
}

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

enum State : uint8_t { RUN, WAIT_WAVE, WAIT_BLOCK, WAIT_GRID, DONE };

struct Lane {
  void *sp = nullptr;
  State st = DONE;
  // pending cross-lane operation
  Op op;
  int width, param;
  uint64_t value, old, result;
  bool bound_ctrl, pred;
  Idx tid;
};

struct Block {
  Lane lanes[MAX_THREADS];
  int n = 0;
  void *sched_sp = nullptr;
  int running = -1;
  const std::function<void()> *body = nullptr;
  char *stacks = nullptr;
  void *lds = nullptr;     // dynamic LDS of this block (a parked block keeps its own)
  Idx bid{0, 0, 0};
};

Block *B = nullptr;

void fiber_main() {
  Block *b = B;
  Lane &me = b->lanes[b->running];
  (*b->body)();
  me.st = DONE;
  emu_switch(&me.sp, b->sched_sp);
  __builtin_trap();   // a finished fiber is never resumed
}

void yield_to_scheduler() {
  Block *b = B;
  Lane &me = b->lanes[b->running];
  emu_switch(&me.sp, b->sched_sp);
  cur.tid = me.tid;   // (the scheduler set the rest)
}

void init_fiber(Block *b, int t) {
  char *top = b->stacks + (size_t)(t + 1) * STACK_BYTES;
  void **sp = (void **)(((uintptr_t)top) & ~(uintptr_t)15);
  sp -= 8;
  for (int k = 0; k < 6; k++) sp[k] = nullptr;   // r15 r14 r13 r12 rbx rbp
  sp[6] = (void *)&fiber_main;                    // return address of the first switch
  sp[7] = nullptr;                                // (keeps rsp = 8 mod 16 at fiber_main's entry, as after a call)
  b->lanes[t].sp = sp;
  b->lanes[t].st = RUN;
}

// result of every lane of a released group
void resolve_group(Block *b, int wave0, int base, int width) {
  const int lo = wave0 + base, hi = wave0 + base + width;
  auto live = [&](int l) { return l >= lo && l < hi && l < b->n && b->lanes[l].st == WAIT_WAVE; };
  // ballot / readfirstlane look at the whole wave's live lanes (they are issued with width 64)
  unsigned long long mask = 0;
  int first = -1;
  for (int l = lo; l < hi && l < b->n; l++)
    if (b->lanes[l].st == WAIT_WAVE) {
      if (b->lanes[l].pred) mask |= 1ull << (l - wave0);
      if (first < 0) first = l;
    }
  for (int l = lo; l < hi && l < b->n; l++) {
    Lane &x = b->lanes[l];
    if (x.st != WAIT_WAVE) continue;
    const int li = l - wave0;   // lane id in the wave
    switch (x.op) {
      case OP_SHFL: {
        const int src = wave0 + base + (x.param & (width - 1));
        x.result = live(src) ? b->lanes[src].value : x.value;
        break;
      }
      case OP_SHFL_XOR: {
        const int src = wave0 + base + (((li - base) ^ x.param) & (width - 1));
        x.result = live(src) ? b->lanes[src].value : x.value;
        break;
      }
      case OP_DPP_SHR1: {
        const int src = l - 1;
        const bool ok = li >= 1 && live(src);
        x.result = ok ? b->lanes[src].value : (x.bound_ctrl ? 0 : x.old);
        break;
      }
      case OP_DPP_SHL1: {
        const int src = l + 1;
        const bool ok = li <= 62 && live(src);
        x.result = ok ? b->lanes[src].value : (x.bound_ctrl ? 0 : x.old);
        break;
      }
      case OP_BALLOT: x.result = mask; break;
      case OP_READFIRST: x.result = b->lanes[first].value; break;
    }
  }
  for (int l = lo; l < hi && l < b->n; l++)
    if (b->lanes[l].st == WAIT_WAVE) b->lanes[l].st = RUN;
}

bool release_waves(Block *b) {
  bool any = false;
  for (int w0 = 0; w0 < b->n; w0 += 64) {
    // narrow groups first
    for (int width = 2; width <= 64; width <<= 1) {
      for (int base = 0; base < 64; base += width) {
        int nwait = 0, nlive = 0;
        bool same = true;
        for (int l = w0 + base; l < w0 + base + width && l < b->n; l++) {
          const Lane &x = b->lanes[l];
          if (x.st == DONE) continue;
          nlive++;
          if (x.st == WAIT_WAVE && x.width == width) nwait++;
          else same = false;
        }
        if (nlive > 0 && same && nwait == nlive) { resolve_group(b, w0, base, width); any = true; }
      }
    }
  }
  return any;
}

// Runs block b until it has finished (true) or until nothing in it can move before another block does: some lane polls
// global memory in emu::grid_yield() and every other live lane waits for it (false: the caller parks the block).
bool run_block(Block *b, bool fresh) {
  B = b;
  cur.bid = b->bid;
  cur.dyn_lds = b->lds;
  if (fresh) for (int t = 0; t < b->n; t++) init_fiber(b, t);
  else for (int t = 0; t < b->n; t++) if (b->lanes[t].st == WAIT_GRID) b->lanes[t].st = RUN;   // poll again
  for (;;) {
    bool progressed = false;
    int ndone = 0;
    for (int t = 0; t < b->n; t++) {
      Lane &x = b->lanes[t];
      if (x.st == DONE) { ndone++; continue; }
      if (x.st != RUN) continue;
      b->running = t;
      cur.tid = x.tid;
      emu_switch(&b->sched_sp, x.sp);
      progressed = true;
      if (x.st == DONE) ndone++;
    }
    if (ndone == b->n) return true;
    if (release_waves(b)) progressed = true;
    {   // __syncthreads: every live thread waits in it
      int nlive = 0, nbar = 0;
      for (int t = 0; t < b->n; t++) {
        if (b->lanes[t].st == DONE) continue;
        nlive++;
        if (b->lanes[t].st == WAIT_BLOCK) nbar++;
      }
      if (nlive > 0 && nbar == nlive) {
        for (int t = 0; t < b->n; t++) if (b->lanes[t].st == WAIT_BLOCK) b->lanes[t].st = RUN;
        progressed = true;
      }
    }
    if (!progressed) {
      for (int t = 0; t < b->n; t++) if (b->lanes[t].st == WAIT_GRID) return false;   // waits for another block
      fprintf(stderr, "[emu] DEADLOCK in block (%u,%u,%u): lanes wait for rendezvous that can never complete\n", cur.bid.x, cur.bid.y, cur.bid.z);
      for (int t = 0; t < b->n; t++) {
        const Lane &x = b->lanes[t];
        if (x.st == DONE) continue;
        fprintf(stderr, "  thread %d: %s op %d width %d\n", t, x.st == WAIT_BLOCK ? "barrier" : "wave-op", (int)x.op, x.width);
        if (t > 80) { fprintf(stderr, "  ...\n"); break; }
      }
      abort();
    }
  }
}

// block contexts: one is enough for an ordinary launch; every parked block holds one until it finishes
std::vector<Block *> pool;
Block *acquire_block() {
  if (!pool.empty()) { Block *b = pool.back(); pool.pop_back(); return b; }
  Block *b = new Block();
  b->stacks = (char *)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (b->stacks == (char *)MAP_FAILED) { perror("[emu] mmap"); abort(); }
  b->lds = std::aligned_alloc(64, 160 * 1024);
  return b;
}

}  // namespace

uint64_t wave_op(Op op, int width, uint64_t value, int param, uint64_t old, bool bound_ctrl, bool pred) {
  Block *b = B;
  Lane &me = b->lanes[b->running];
  if (op != OP_SHFL && op != OP_SHFL_XOR) width = 64;
  if (width <= 1) return value;
  me.op = op; me.width = width; me.value = value; me.param = param; me.old = old; me.bound_ctrl = bound_ctrl; me.pred = pred;
  me.st = WAIT_WAVE;
  yield_to_scheduler();
  return me.result;
}

void block_barrier() {
  Block *b = B;
  Lane &me = b->lanes[b->running];
  me.st = WAIT_BLOCK;
  yield_to_scheduler();
}

// the polling lane of a grid-level wait: let the other blocks run (the GPU code sleeps here)
void grid_yield() {
  Block *b = B;
  Lane &me = b->lanes[b->running];
  me.st = WAIT_GRID;
  yield_to_scheduler();
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
  static std::mutex mu;   // host threads of dada2hip_run_multi: one emulated launch at a time
  std::lock_guard<std::mutex> guard(mu);
  if (B) { fprintf(stderr, "[emu] nested launch\n"); abort(); }
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > (unsigned)MAX_THREADS || lds_bytes > 160 * 1024) { fprintf(stderr, "[emu] bad launch geometry\n"); abort(); }
  cur.bdim = Idx{block.x, block.y, block.z};
  cur.gdim = Idx{grid.x, grid.y, grid.z};
  std::vector<Block *> parked;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        Block *blk = acquire_block();
        blk->n = (int)nthreads;
        blk->body = &body;
        blk->bid = Idx{bx, by, bz};
        for (unsigned t = 0; t < nthreads; t++) blk->lanes[t].tid = Idx{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        if (run_block(blk, true)) pool.push_back(blk);
        else {
          parked.push_back(blk);
          if (parked.size() > 64) { fprintf(stderr, "[emu] more than 64 blocks wait for one another: not a grid this emulator holds\n"); abort(); }
        }
      }
  while (!parked.empty())
    for (size_t k = 0; k < parked.size();) {
      if (run_block(parked[k], false)) { pool.push_back(parked[k]); parked[k] = parked.back(); parked.pop_back(); }
      else k++;
    }
  B = nullptr;
}

}  // namespace emu
