// tests/hostsim/hipsim_rt.cpp -- TEST INFRASTRUCTURE ONLY: the fiber scheduler behind
// tests/hostsim/hip/hip_runtime.h (see there).  One block at a time; every lane of the block is a
// fiber (its own stack; a register-saving switch on x86-64, ucontext elsewhere).  Lanes run until they park at a cross-lane operation or finish.  When nothing in the
// block can run:
//   * per wavefront, the parked lanes that are furthest "behind" go: deepest calling frame first
//     (stacks grow down), then lowest source position of the site (the (line, column) of every inlined
//     frame, outermost first, ranked at build time from the debug info).  In source order a divergent
//     branch or a loop body comes before the code that follows it: the lanes inside go first and the
//     others wait for them at the join, which is what the hardware's reconvergence does.  Shuffles read the value the source lane brought (a
//     source outside the meeting yields the reader's own value -- undefined on hardware), ballots /
//     all / any see the meeting's lanes only;
//   * if no wavefront had anything to meet over, every live lane must stand at __syncthreads, and the
//     barrier opens; anything else is a deadlock and aborts.
// A meeting that leaves live lanes of the wavefront elsewhere (divergent use of a wave operation) is
// counted in hipsim_partial_wave_ops().
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <sys/mman.h>

#include <algorithm>
#include <map>
#include <string>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <vector>

#include <hip/hip_runtime.h>

thread_local hipsim_idx blockIdx, threadIdx, blockDim, gridDim;

namespace hipsim {
namespace {
enum { RUN = 0, WAVE, BARRIER, DONE, MEM };
#if defined(__x86_64__)
// a context is a stack pointer: the callee-saved registers live on the stack it points at
// (swapcontext would also make a system call per switch for the signal mask)
typedef void *Ctx;
extern "C" void hipsim_switch(Ctx *save, Ctx to);
asm(R"(
    .text
    .globl hipsim_switch
    .type hipsim_switch,@function
hipsim_switch:
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
    .size hipsim_switch, .-hipsim_switch
)");
static void ctx_make(Ctx *c, char *stack, size_t size, void (*fn)()) {
    uintptr_t top = ((uintptr_t)(stack + size) & ~(uintptr_t)15) - 32;   // 16-aligned slot for the entry address
    void **sp = (void **)top;
    sp[0] = (void *)fn;                                                   // `ret` lands in fn with rsp % 16 == 8
    sp[1] = nullptr;
    sp -= 6;                                                              // r15 r14 r13 r12 rbx rbp
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    *c = (Ctx)sp;
}
static inline void ctx_switch(Ctx *save, Ctx *to) { hipsim_switch(save, *to); }
#else
typedef ucontext_t Ctx;
static void ctx_make(Ctx *c, char *stack, size_t size, void (*fn)()) {
    getcontext(c);
    c->uc_stack.ss_sp = stack;
    c->uc_stack.ss_size = size;
    c->uc_link = nullptr;
    makecontext(c, fn, 0);
}
static inline void ctx_switch(Ctx *save, Ctx *to) { swapcontext(save, to); }
#endif

struct Fiber {
    Ctx ctx;
    int state, kind, arg;
    const void *site, *frame;
    uint64_t rank;                 // program-order rank of `site` (sites file), else its address
    uint64_t val, res;
};
constexpr size_t STACK = 256 * 1024, MAX_THREADS = 1024;
thread_local std::vector<Fiber> F;
thread_local Ctx sched;
thread_local int cur;
thread_local const std::function<void()> *body;
thread_local char *stacks;
thread_local std::vector<char> dyn;
thread_local bool in_fiber;
unsigned long partial_ops;
thread_local const char *host_lo, *host_hi;        // the launching thread's stack (kernel arguments are read from it)
unsigned long long mem_bytes[2], mem_ops[2];     // [load, store]: kernel accesses outside the lanes' stacks

// program-order ranks of the parking sites, made at build time from the debug info (tests/hostsim/__init__.py)
std::vector<std::pair<uint64_t, uint64_t>> site_rank;
uintptr_t lib_base;
void load_sites() {
    Dl_info di;
    if (!dladdr((void *)&load_sites, &di) || !di.dli_fname) return;
    lib_base = (uintptr_t)di.dli_fbase;
    FILE *f = fopen((std::string(di.dli_fname) + ".sites").c_str(), "rb");
    if (!f) return;
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) == 1) {
        site_rank.resize(n);
        if (fread(site_rank.data(), 16, n, f) != n) site_rank.clear();
    }
    fclose(f);
}
uint64_t rank_of(const void *site) {
    const uint64_t off = (uintptr_t)site - lib_base;
    auto it = std::lower_bound(site_rank.begin(), site_rank.end(), std::make_pair(off, (uint64_t)0));
    if (it != site_rank.end() && it->first == off) return it->second;
    return (1ull << 40) + off;                                    // unknown site: behind every known one, by address
}

void trampoline() {
    (*body)();
    F[cur].state = DONE;
    ctx_switch(&F[cur].ctx, &sched);
    abort();
}

void run_block(unsigned n) {
    F.resize(n);
    for (unsigned i = 0; i < n; i++) {
        Fiber &f = F[i];
        ctx_make(&f.ctx, stacks + (size_t)i * STACK, STACK, trampoline);
        f.state = RUN;
    }
    for (unsigned long rounds = 0;; rounds++) {
        if (rounds > 20000000ul) {
            fprintf(stderr, "hipsim: block (%u,%u) does not finish (livelock?)\n", blockIdx.x, blockIdx.y);
            for (unsigned i = 0; i < n; i++)
                fprintf(stderr, "  lane %u state %d kind %d site %p frame %p val %llu arg %d\n", i, F[i].state, F[i].kind, F[i].site,
                        F[i].frame, (unsigned long long)F[i].val, F[i].arg);
            abort();
        }
        for (unsigned i = 0; i < n; i++)
            if (F[i].state == RUN) {
                cur = (int)i;
                threadIdx = {i, 0, 0};
                in_fiber = true;
                ctx_switch(&sched, &F[i].ctx);
                in_fiber = false;
            }
        unsigned live = 0, at_barrier = 0;
        for (unsigned i = 0; i < n; i++) { live += F[i].state != DONE; at_barrier += F[i].state == BARRIER; }
        if (!live) return;
        bool met = false;
        for (unsigned w0 = 0; w0 < n; w0 += 64) {
            const unsigned w1 = w0 + 64 < n ? w0 + 64 : n;
            // memory accesses first: a lane that can still move on its own is not at a meeting point yet, and a
            // cross-lane operation only meets once every live lane of the wavefront stands at one (or at the
            // barrier).  Among candidates, the lane furthest "behind" leads.
            int lead = -1;
            for (int pass = 0; pass < 2 && lead < 0; pass++)
                for (unsigned i = w0; i < w1; i++) {
                    if (F[i].state != (pass == 0 ? MEM : WAVE)) continue;
                    if (lead < 0) { lead = (int)i; continue; }
                    const uintptr_t fi = (uintptr_t)F[i].frame, fl = (uintptr_t)F[lead].frame;
                    if (fi < fl || (fi == fl && F[i].rank < F[lead].rank)) lead = (int)i;
                }
            if (lead < 0) continue;
            const Fiber &L = F[lead];
            if (L.state == MEM) {                                   // a memory access: the lanes at this instruction do it now
                for (unsigned i = w0; i < w1; i++)
                    if (F[i].state == MEM && F[i].site == L.site) F[i].state = RUN;
                met = true;
                continue;
            }
            uint64_t in = 0, ballot = 0;
            bool elsewhere = false;
            for (unsigned i = w0; i < w1; i++) {
                if (F[i].state == WAVE && F[i].site == L.site && F[i].kind == L.kind) {
                    in |= 1ull << (i - w0);
                    if (F[i].val) ballot |= 1ull << (i - w0);
                } else if (F[i].state != DONE) elsewhere = true;
            }
            if (elsewhere) {
                partial_ops++;
                if (getenv("HIPSIM_TRACE"))
                    fprintf(stderr, "hipsim: partial meeting block (%u,%u) wave %u kind %d site %p frame %p lanes %016llx\n", blockIdx.x,
                            blockIdx.y, w0 / 64, L.kind, L.site, L.frame, (unsigned long long)in);
            }
            for (unsigned i = w0; i < w1; i++) {
                if (!((in >> (i - w0)) & 1)) continue;
                Fiber &f = F[i];
                switch (f.kind) {
                    case K_SHFL: f.res = ((in >> f.arg) & 1) ? F[w0 + (unsigned)f.arg].val : f.val; break;
                    case K_BALLOT: f.res = ballot; break;
                    case K_ALL: f.res = ballot == in; break;
                    case K_ANY: f.res = ballot != 0; break;
                }
                f.state = RUN;
            }
            met = true;
        }
        if (met) continue;
        if (at_barrier != live) {
            fprintf(stderr, "hipsim: deadlock in block %u: %u live lanes, %u at the barrier\n", blockIdx.x, live, at_barrier);
            abort();
        }
        for (unsigned i = 0; i < n; i++) if (F[i].state == BARRIER) F[i].state = RUN;
    }
}
}  // namespace

uint64_t collective(int kind, const void *site, const void *frame, uint64_t val, int arg) {
    Fiber &f = F[cur];
    f.kind = kind; f.site = site; f.rank = rank_of(site);
    f.frame = (const void *)((uintptr_t)frame - (uintptr_t)(stacks + (size_t)cur * STACK));   // offset in the lane's own stack
    f.val = val; f.arg = arg;
    f.state = kind == K_BARRIER ? BARRIER : WAVE;
    ctx_switch(&f.ctx, &sched);
    return F[cur].res;
}

std::map<uintptr_t, uintptr_t> poisoned;          // start -> end
std::vector<std::pair<uintptr_t, uintptr_t>> poison_flat;
bool poison_dirty;
std::map<uintptr_t, size_t> blocks;               // hipMalloc'd blocks

// A lane is about to touch memory that is not its own stack (the library is compiled with
// -fsanitize=thread only to get this call in front of every such load and store): it parks, and goes on
// when it is its instruction's turn -- the lanes of a wavefront advance through their memory accesses in
// program order like the lock-step hardware, so "every lane reads X, then lane 0 overwrites X" keeps its
// meaning even though the lanes are run one after the other.
void before_access(const void *addr, size_t size, int is_store, const void *site, const void *frame) {
    if (!in_fiber) return;
    const char *lo = stacks + (size_t)cur * STACK;
    if ((const char *)addr >= lo && (const char *)addr < lo + STACK) return;
    if ((const char *)addr < host_lo || (const char *)addr >= host_hi) {   // not the launcher's stack: kernel arguments live there
        mem_bytes[is_store] += size; mem_ops[is_store]++;
    }
    if (!poisoned.empty()) {                                      // [start, end) ranges, disjoint, by start
        const uintptr_t a = (uintptr_t)addr;
        if (poison_dirty) {
            poison_flat.assign(poisoned.begin(), poisoned.end());
            poison_dirty = false;
        }
        auto it = std::upper_bound(poison_flat.begin(), poison_flat.end(), std::make_pair(a + size - 1, ~(uintptr_t)0));
        if (it != poison_flat.begin() && (--it)->second > a) {
            fprintf(stderr, "hipsim: kernel access of %zu bytes at %p runs into an arena guard gap [%p, %p): site +0x%lx, block (%u,%u) "
                    "thread %u\n", size, addr, (void *)it->first, (void *)it->second, (unsigned long)((uintptr_t)site - lib_base),
                    blockIdx.x, blockIdx.y, threadIdx.x);
            abort();
        }
    }
    Fiber &f = F[cur];
    f.site = site; f.rank = rank_of(site);
    f.frame = (const void *)((uintptr_t)frame - (uintptr_t)lo);
    f.state = MEM;
    ctx_switch(&f.ctx, &sched);
}

void *dyn_shared() { return dyn.data(); }

void poison(const void *p, size_t n) { if (n) { poisoned[(uintptr_t)p] = (uintptr_t)p + n; poison_dirty = true; } }
void *dev_malloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (p) blocks[(uintptr_t)p] = n;
    return p;
}
void dev_free(void *p) {
    auto b = blocks.find((uintptr_t)p);
    if (b != blocks.end()) {
        poisoned.erase(poisoned.lower_bound(b->first), poisoned.lower_bound(b->first + b->second));
        poison_dirty = true;
        blocks.erase(b);
    }
    free(p);
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &fn) {
    const unsigned n = block.x * block.y * block.z;
    if (n == 0 || n > MAX_THREADS || block.y != 1 || block.z != 1) {
        fprintf(stderr, "hipsim: unsupported launch shape\n");
        abort();
    }
    if (!host_lo) {
        pthread_attr_t at;
        void *sa = nullptr; size_t sz = 0;
        if (!pthread_getattr_np(pthread_self(), &at)) { pthread_attr_getstack(&at, &sa, &sz); pthread_attr_destroy(&at); }
        host_lo = (const char *)sa; host_hi = host_lo + sz;
    }
    if (!stacks) {
        if (site_rank.empty()) load_sites();
        stacks = (char *)mmap(nullptr, STACK * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == (char *)MAP_FAILED) { perror("hipsim: mmap"); abort(); }
    }
    if (dyn.size() < shmem + 64) dyn.resize(shmem + 64);
    body = &fn;
    gridDim = {grid.x, grid.y, grid.z}; blockDim = {block.x, 1, 1};
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blockIdx = {bx, by, bz};
                run_block(n);
            }
}
}  // namespace hipsim

extern "C" unsigned long hipsim_partial_wave_ops(void) { return hipsim::partial_ops; }
// bytes / accesses the kernels made to memory outside the lanes' stacks since the last reset: out[0..3] =
// bytes loaded, bytes stored, loads, stores (an upper bound of a kernel's algorithmic traffic: no caches, no merging)
extern "C" void hipsim_traffic(unsigned long long *out, int reset) {
    out[0] = hipsim::mem_bytes[0]; out[1] = hipsim::mem_bytes[1]; out[2] = hipsim::mem_ops[0]; out[3] = hipsim::mem_ops[1];
    if (reset) hipsim::mem_bytes[0] = hipsim::mem_bytes[1] = hipsim::mem_ops[0] = hipsim::mem_ops[1] = 0;
}

// the -fsanitize=thread hooks (no ThreadSanitizer runtime is linked; these are all there is)
#define HIPSIM_HOOK(name, size, st)                                                                             \
    extern "C" __attribute__((noinline)) void name(void *a) {                                                \
        hipsim::before_access(a, size, st, __builtin_return_address(0), __builtin_frame_address(1));      \
    }
HIPSIM_HOOK(__tsan_read1, 1, 0) HIPSIM_HOOK(__tsan_read2, 2, 0) HIPSIM_HOOK(__tsan_read4, 4, 0) HIPSIM_HOOK(__tsan_read8, 8, 0) HIPSIM_HOOK(__tsan_read16, 16, 0)
HIPSIM_HOOK(__tsan_write1, 1, 1) HIPSIM_HOOK(__tsan_write2, 2, 1) HIPSIM_HOOK(__tsan_write4, 4, 1) HIPSIM_HOOK(__tsan_write8, 8, 1) HIPSIM_HOOK(__tsan_write16, 16, 1)
HIPSIM_HOOK(__tsan_unaligned_read2, 2, 0) HIPSIM_HOOK(__tsan_unaligned_read4, 4, 0) HIPSIM_HOOK(__tsan_unaligned_read8, 8, 0) HIPSIM_HOOK(__tsan_unaligned_read16, 16, 0)
HIPSIM_HOOK(__tsan_unaligned_write2, 2, 1) HIPSIM_HOOK(__tsan_unaligned_write4, 4, 1) HIPSIM_HOOK(__tsan_unaligned_write8, 8, 1) HIPSIM_HOOK(__tsan_unaligned_write16, 16, 1)
extern "C" void __tsan_init(void) {}
extern "C" void __tsan_func_entry(void *) {}
extern "C" void __tsan_func_exit(void) {}
extern "C" void __tsan_vptr_update(void **, void *) {}
extern "C" void __tsan_vptr_read(void **) {}
extern "C" void __tsan_read_range(void *, unsigned long) {}
extern "C" void __tsan_write_range(void *, unsigned long) {}
