// Runtime of the HIP-on-CPU emulation (see include/hip/hip_runtime.h): fibers + wavefront rendezvous + a synchronous subset of the HIP API.
// Test infrastructure only.
#include <hip/hip_runtime.h>

#include <chrono>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <set>
#include <sys/mman.h>
#include <unistd.h>

namespace hipcpu {

thread_local uint3 t_threadIdx, t_blockIdx;
thread_local dim3 t_blockDim, t_gridDim;

// ---- context switch (x86-64 SysV: callee-saved registers on the fiber's own stack) ---------------------------------------------------
extern "C" void hipcpu_switch(void** save_sp, void* load_sp);
#if defined(__x86_64__)
asm(".text\n"
    ".globl hipcpu_switch\n"
    ".type hipcpu_switch,@function\n"
    "hipcpu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size hipcpu_switch,.-hipcpu_switch\n");
#else
#error "hipcpu: the fiber switch is written for x86-64"
#endif

// AddressSanitizer keeps the bounds of the stack it believes the thread is on: tell it about every switch (the kernels can then be run under
// -fsanitize=address, which the GPU pool refuses on the device)
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPCPU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#endif

enum { ST_RUN = 0, ST_WAVE, ST_BLOCK, ST_DONE };
static const size_t STACK_BYTES = 512 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int st = ST_DONE;
    unsigned tx = 0, ty = 0, tz = 0;
    int lane = 0, wave = 0;
    int op = 0;
    const void* site = nullptr;
    OpArgs a{};
    OpOut o{};
    int pred = 0;
    void* asan_fake = nullptr;
};

struct Worker {
    std::vector<Fiber*> pool;
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    void* dyn = nullptr;
    size_t dyn_cap = 0;
    bool in_kernel = false;
    std::vector<void*> poisoned;            // static __shared__ objects already filled for the running workgroup
    const void* sched_stack = nullptr;      // (ASan) the scheduler's own stack, learnt at the first switch into a fiber
    size_t sched_stack_size = 0;
    ~Worker()
    {
        for (Fiber* f : pool) { if (f->stack) munmap(f->stack - 4096, STACK_BYTES + 4096); delete f; }
        free(dyn);
    }
};
static thread_local Worker W;

static struct Stats { unsigned long long kernels = 0, blocks = 0, wave_ops = 0, barriers = 0, readlane_inactive = 0, shfl_inactive = 0; } g_stats;

static void fiber_main()
{
#ifdef HIPCPU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &W.sched_stack, &W.sched_stack_size);
#endif
    (*W.body)();
    W.cur->st = ST_DONE;
#ifdef HIPCPU_ASAN
    __sanitizer_start_switch_fiber(nullptr, W.sched_stack, W.sched_stack_size);      // null: this fiber's stack is done with
#endif
    hipcpu_switch(&W.cur->sp, W.sched_sp);
    abort();      // a finished fiber is never resumed
}

static Fiber* fiber_at(size_t i)
{
    while (W.pool.size() <= i) {
        Fiber* f = new Fiber;
        char* m = (char*)mmap(nullptr, STACK_BYTES + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == (char*)MAP_FAILED) { fprintf(stderr, "hipcpu: cannot map a fiber stack\n"); abort(); }
        mprotect(m, 4096, PROT_NONE);                 // guard page under the stack
        f->stack = m + 4096;
        W.pool.push_back(f);
    }
    return W.pool[i];
}

static void fiber_reset(Fiber* f)
{
    uintptr_t top = ((uintptr_t)f->stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                                   // return address of fiber_main (never used)
    *--sp = (void*)fiber_main;
    for (int i = 0; i < 6; i++) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
    f->sp = sp;
    f->st = ST_RUN;
}

static inline void run_fiber(Fiber* f)
{
    W.cur = f;
    t_threadIdx = uint3{f->tx, f->ty, f->tz};
#ifdef HIPCPU_ASAN
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, f->stack, STACK_BYTES);
#endif
    hipcpu_switch(&W.sched_sp, f->sp);
#ifdef HIPCPU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

static inline void yield_to_scheduler()
{
    Fiber* f = W.cur;
#ifdef HIPCPU_ASAN
    __sanitizer_start_switch_fiber(&f->asan_fake, W.sched_stack, W.sched_stack_size);
#endif
    hipcpu_switch(&f->sp, W.sched_sp);
    // resumed: the scheduler has set t_threadIdx and W.cur for us
#ifdef HIPCPU_ASAN
    __sanitizer_finish_switch_fiber(f->asan_fake, nullptr, nullptr);
#endif
}

int lane_id() { return W.cur->lane; }
void* dyn_lds() { return W.dyn; }
void poison_static_lds(void* p, size_t bytes)
{
    for (void* q : W.poisoned) if (q == p) return;
    W.poisoned.push_back(p);
    memset(p, 0xFF, bytes);
}
long long wall_clock()
{
    using namespace std::chrono;
    return duration_cast<nanoseconds>(steady_clock::now().time_since_epoch()).count() / 10;      // 100 MHz, the rate hipDeviceAttributeWallClockRate reports
}

OpOut wave_op(int op, const OpArgs& a)
{
    if (!W.in_kernel) { fprintf(stderr, "hipcpu: cross-lane operation outside a kernel\n"); abort(); }
    Fiber* f = W.cur;
    f->op = op;
    f->site = __builtin_extract_return_addr(__builtin_return_address(0));
    f->a = a;
    f->st = ST_WAVE;
    yield_to_scheduler();
    return f->o;
}

void block_barrier()
{
    Fiber* f = W.cur;
    f->pred = 0;
    f->st = ST_BLOCK;
    yield_to_scheduler();
}

int block_barrier_or(int pred)
{
    Fiber* f = W.cur;
    f->pred = pred ? 1 : 0;
    f->st = ST_BLOCK;
    yield_to_scheduler();
    return f->pred;
}

// the lanes of one wavefront that wait at the same (site, op): their results from each other's operands
static void resolve_group(Fiber** lane /* [64], null where no such work-item */, unsigned long long mask)
{
    const int op = lane[__builtin_ctzll(mask)]->op;
    auto active = [&](int j) { return j >= 0 && j < 64 && ((mask >> j) & 1ull); };
    unsigned long long ballot = 0;
    if (op == OP_BALLOT)
        for (int i = 0; i < 64; i++) if (active(i) && lane[i]->a.in0) ballot |= 1ull << i;
    const int first = __builtin_ctzll(mask);
    for (int i = 0; i < 64; i++) {
        if (!active(i)) continue;
        Fiber* f = lane[i];
        const OpArgs& a = f->a;
        OpOut o{0, 0, mask, mask};
        switch (op) {
        case OP_WAVE_BARRIER: break;
        case OP_SHFL_IDX: {
            const int s = a.p0 & 63;
            if (active(s)) { o.out0 = lane[s]->a.in0; o.out1 = lane[s]->a.in1; }
            else { o.out0 = a.in0; o.out1 = a.in1; g_stats.shfl_inactive++; }
            break;
        }
        case OP_BALLOT: o.mask = ballot; break;
        case OP_READLANE: {
            const int s = a.p0 & 63;
            if (active(s)) o.out0 = lane[s]->a.in0; else { o.out0 = a.in0; g_stats.readlane_inactive++; }
            break;
        }
        case OP_READFIRST: o.out0 = lane[first]->a.in0; break;
        case OP_SWAP32: {      // v_permlane32_swap: lanes 32..63 of vdst <-> lanes 0..31 of vsrc   (vdst = old0, vsrc = in0)
            o.out0 = (i >= 32 && active(i - 32)) ? lane[i - 32]->a.in0 : a.old0;
            o.out1 = (i < 32 && active(i + 32)) ? lane[i + 32]->a.old0 : a.in0;
            break;
        }
        case OP_SWAP16: {      // v_permlane16_swap: odd rows of vdst <-> even rows of vsrc
            const int r = i >> 4;
            o.out0 = ((r & 1) && active(i - 16)) ? lane[i - 16]->a.in0 : a.old0;
            o.out1 = (!(r & 1) && active(i + 16)) ? lane[i + 16]->a.old0 : a.in0;
            break;
        }
        case OP_DPP: {
            const int ctrl = a.p0, row_mask = a.p1, bank_mask = a.p2, bound = a.p3;
            const int r = i >> 4, pos = i & 15, bank = pos >> 2;
            if (!((row_mask >> r) & 1) || !((bank_mask >> bank) & 1)) { o.out0 = a.old0; break; }
            int src = -1;                                                   // lane index, -1 = out of range
            if (ctrl >= 0 && ctrl <= 0xff) src = (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);            // quad_perm
            else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int p = pos + (ctrl & 15); if (p < 16) src = r * 16 + p; }     // row_shl
            else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int p = pos - (ctrl & 15); if (p >= 0) src = r * 16 + p; }     // row_shr
            else if (ctrl >= 0x121 && ctrl <= 0x12f) src = r * 16 + ((pos - (ctrl & 15)) & 15);                             // row_ror
            else if (ctrl == 0x140) src = r * 16 + (15 - pos);                                                              // row_mirror
            else if (ctrl == 0x141) src = r * 16 + ((pos & 8) | (7 - (pos & 7)));                                            // row_half_mirror
            else if (ctrl == 0x142) { if (r >= 1) src = r * 16 - 1; }                                                       // row_bcast15
            else if (ctrl == 0x143) { if (r >= 2) src = 31; }                                                               // row_bcast31
            else { fprintf(stderr, "hipcpu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
            if (src >= 0 && active(src)) o.out0 = lane[src]->a.in0;
            else o.out0 = bound ? 0u : a.old0;
            break;
        }
        default: fprintf(stderr, "hipcpu: unknown wave operation %d\n", op); abort();
        }
        f->o = o;
        f->st = ST_RUN;
    }
    g_stats.wave_ops++;
}

enum { SCHED_NATURAL = 0, SCHED_REVERSE, SCHED_SHUFFLE };
static int g_sched_mode = -1;
static unsigned g_sched_seed = 1;
static std::mt19937 g_rng;

static void schedule_init()
{
    if (g_sched_mode >= 0) return;
    g_sched_mode = SCHED_NATURAL;
    if (const char* e = getenv("HIPCPU_SCHEDULE")) {
        if (!strncmp(e, "reverse", 7)) g_sched_mode = SCHED_REVERSE;
        else if (!strncmp(e, "shuffle", 7)) { g_sched_mode = SCHED_SHUFFLE; if (e[7] == ':') g_sched_seed = (unsigned)atoi(e + 8); g_rng.seed(g_sched_seed); }
    }
}

static void order_of(int n, std::vector<int>& idx)
{
    idx.resize(n);
    for (int i = 0; i < n; i++) idx[i] = g_sched_mode == SCHED_REVERSE ? n - 1 - i : i;
    if (g_sched_mode == SCHED_SHUFFLE) std::shuffle(idx.begin(), idx.end(), g_rng);
}

static void run_block(const std::function<void()>& body, dim3 block)
{
    const int n = (int)(block.x * block.y * block.z);
    const int nw = (n + 63) / 64;
    for (int t = 0; t < n; t++) {
        Fiber* f = fiber_at((size_t)t);
        fiber_reset(f);
        f->tx = (unsigned)t % block.x; f->ty = ((unsigned)t / block.x) % block.y; f->tz = (unsigned)t / (block.x * block.y);
        f->lane = t & 63; f->wave = t >> 6;
    }
    W.body = &body;
    std::vector<int> wave_order, lane_order;
    for (;;) {
        order_of(nw, wave_order);
        for (int wi = 0; wi < nw; wi++) {
            const int w = wave_order[wi];
            const int lanes = std::min(64, n - w * 64);
            Fiber* lane[64];
            for (int l = 0; l < 64; l++) lane[l] = l < lanes ? W.pool[(size_t)w * 64 + l] : nullptr;
            for (;;) {
                order_of(lanes, lane_order);
                for (int li = 0; li < lanes; li++) {
                    Fiber* f = lane[lane_order[li]];
                    if (f->st == ST_RUN) run_fiber(f);               // until it blocks or finishes
                }
                // no runnable lane is left in this wave: release the lanes that wait at cross-lane operations, one (site, op) group at a time
                bool released = false;
                unsigned long long todo = 0;
                for (int l = 0; l < lanes; l++) if (lane[l]->st == ST_WAVE) todo |= 1ull << l;
                while (todo) {
                    const int l0 = __builtin_ctzll(todo);
                    unsigned long long mask = 0;
                    for (int l = l0; l < lanes; l++)
                        if (((todo >> l) & 1ull) && lane[l]->site == lane[l0]->site && lane[l]->op == lane[l0]->op) mask |= 1ull << l;
                    resolve_group(lane, mask);
                    todo &= ~mask;
                    released = true;
                }
                if (!released) break;                                // every lane of the wave is at the workgroup barrier or done
            }
        }
        // every work-item of the workgroup is at __syncthreads or done
        int waiting = 0, any = 0;
        for (int t = 0; t < n; t++) if (W.pool[(size_t)t]->st == ST_BLOCK) { waiting++; any |= W.pool[(size_t)t]->pred; }
        if (!waiting) break;
        for (int t = 0; t < n; t++) if (W.pool[(size_t)t]->st == ST_BLOCK) { W.pool[(size_t)t]->pred = any; W.pool[(size_t)t]->st = ST_RUN; }
        g_stats.barriers++;
    }
    W.body = nullptr;
}

static std::mutex g_kernel_mutex;                           // one kernel at a time process-wide: `__shared__` is a function-local static
static std::mutex g_attr_mutex;
static std::map<const void*, size_t> g_dyn_lds_limit;      // hipFuncAttributeMaxDynamicSharedMemorySize per kernel
static thread_local hipError_t t_last_error = hipSuccess;
static const size_t LDS_PER_BLOCK = 160 * 1024;

static int emulated_cus()
{
    static const int n = []() { const char* e = getenv("HIPCPU_CUS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();      // the launch geometry of the real part (256 CUs = 8 XCDs x 32) unless told otherwise
    return n;
}

static void run_kernel(dim3 grid, dim3 block, size_t dyn, const std::function<void()>& body, const char* name)
{
    std::lock_guard<std::mutex> lk(g_kernel_mutex);
    schedule_init();
    if (W.in_kernel) { fprintf(stderr, "hipcpu: nested kernel launch\n"); abort(); }
#ifdef HIPCPU_ASAN
    // exactly the bytes the launch asked for, so that a kernel reading or writing past its dynamic LDS runs into the sanitizer's red zone
    free(W.dyn); W.dyn = nullptr; W.dyn_cap = 0;
    if (dyn) { if (posix_memalign(&W.dyn, 64, dyn) != 0) abort(); W.dyn_cap = dyn; }
#else
    if (dyn > W.dyn_cap) { free(W.dyn); W.dyn = aligned_alloc(64, (dyn + 63) & ~(size_t)63); W.dyn_cap = dyn; }
#endif
    if (getenv("HIPCPU_TRACE")) fprintf(stderr, "[hipcpu] %s grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, dyn);
    W.in_kernel = true;
    t_blockDim = block; t_gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                t_blockIdx = uint3{bx, by, bz};
                if (dyn) memset(W.dyn, 0xFF, dyn);                // LDS is not zeroed between workgroups: poison it (NaN as a float, -1 as an int)
                W.poisoned.clear();
                run_block(body, block);
                g_stats.blocks++;
            }
    W.in_kernel = false;
    g_stats.kernels++;
}

// ---- streams, events, graphs ---------------------------------------------------------------------------------------------------------------
struct Stream { std::vector<std::function<void()>>* capture = nullptr; };
struct Event { std::chrono::steady_clock::time_point t; bool recorded = false; };
struct Graph { std::vector<std::function<void()>> ops; };
struct GraphExec { std::vector<std::function<void()>> ops; };
static Stream g_null_stream;
static inline Stream* S(hipStream_t s) { return s ? s : &g_null_stream; }
static inline void submit(hipStream_t s, std::function<void()> fn)
{
    Stream* st = S(s);
    if (st->capture) st->capture->push_back(std::move(fn)); else fn();
}

void launch_closure(dim3 grid, dim3 block, size_t dyn, hipStream_t s, std::function<void()> body, const char* name, const void* kernel)
{
    const size_t threads = (size_t)block.x * block.y * block.z;
    size_t limit = 64 * 1024;
    { std::lock_guard<std::mutex> lk(g_attr_mutex); auto it = g_dyn_lds_limit.find(kernel); if (it != g_dyn_lds_limit.end()) limit = std::max(limit, it->second); }
    if (threads == 0 || threads > 1024 || grid.x == 0 || grid.y == 0 || grid.z == 0 || dyn > limit || dyn > LDS_PER_BLOCK) {
        // what the device does with such a launch: nothing runs, the error is reported by hipGetLastError
        fprintf(stderr, "hipcpu: launch of %s refused: grid (%u,%u,%u) block (%u,%u,%u) dynamic LDS %zu (limit %zu)\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, dyn, limit);
        t_last_error = hipErrorInvalidValue;
        return;
    }
    std::string nm(name);
    if (getenv("HIPCPU_LAUNCH_LOG")) {            // which kernel INSTANTIATIONS a run launches: the symbol behind the function pointer (tools/kernel_launch_coverage.py)
        static std::mutex m;
        static std::set<std::string> seen;
        Dl_info di;
        if (dladdr(kernel, &di) && di.dli_sname) {
            std::lock_guard<std::mutex> lk(m);
            if (seen.insert(di.dli_sname).second) {
                char path[512];
                snprintf(path, sizeof(path), "%s.%d", getenv("HIPCPU_LAUNCH_LOG"), (int)getpid());
                if (FILE* fh = fopen(path, "a")) { fprintf(fh, "%s\n", di.dli_sname); fclose(fh); }
            }
        }
    }
    submit(s, [grid, block, dyn, body, nm]() { run_kernel(grid, block, dyn, body, nm.c_str()); });
}

}  // namespace hipcpu

using namespace hipcpu;

// HIPCPU_STATS=1: the counters on stderr when the process ends (how many cross-lane operations read a lane outside the exec mask, for instance)
static struct StatsAtExit {
    ~StatsAtExit()
    {
        if (getenv("HIPCPU_STATS"))
            fprintf(stderr, "[hipcpu] kernels %llu workgroups %llu cross-lane ops %llu barriers %llu readlane-of-inactive-lane %llu shuffle-from-inactive-lane %llu\n",
                    g_stats.kernels, g_stats.blocks, g_stats.wave_ops, g_stats.barriers, g_stats.readlane_inactive, g_stats.shfl_inactive);
    }
} g_stats_at_exit;

extern "C" void hipcpu_stats(unsigned long long* out6)
{
    out6[0] = g_stats.kernels; out6[1] = g_stats.blocks; out6[2] = g_stats.wave_ops; out6[3] = g_stats.barriers; out6[4] = g_stats.readlane_inactive; out6[5] = g_stats.shfl_inactive;
}

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev)
{
    if (dev != 0) return hipErrorInvalidDevice;
    switch (a) {
    case hipDeviceAttributeMultiprocessorCount: *v = emulated_cus(); return hipSuccess;
    case hipDeviceAttributeWallClockRate: *v = 100000; return hipSuccess;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: { const char* e = getenv("HIPCPU_LDS_ATTR"); *v = e ? atoi(e) : (int)LDS_PER_BLOCK; return hipSuccess; }     // (a runtime may report the 64 KB a kernel gets without the opt-in)
    }
    return hipErrorInvalidValue;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev)
{
    if (dev != 0) return hipErrorInvalidDevice;
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "hipcpu emulation (no GPU)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:emulated-on-cpu");
    p->multiProcessorCount = emulated_cus();
    p->totalGlobalMem = (size_t)16 << 30;
    p->sharedMemPerBlock = 64 * 1024;
    return hipSuccess;
}
hipError_t hipGetLastError() { const hipError_t e = t_last_error; t_last_error = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNoDevice: return "no ROCm-capable device is detected";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorNotReady: return "device not ready";
    default: return "unknown error";
    }
}
static std::mutex g_mem_mutex;
static std::map<void*, size_t> g_device_allocs;                       // "device" memory in use (what a leak test asks torch.cuda.mem_get_info for)
static size_t g_device_bytes = 0;
extern "C" unsigned long long hipcpu_device_bytes_in_use() { std::lock_guard<std::mutex> lk(g_mem_mutex); return g_device_bytes; }
hipError_t hipMalloc(void** p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) return hipSuccess;
    if (bytes > ((size_t)48 << 30)) return hipErrorOutOfMemory;
    void* m = nullptr;
    if (posix_memalign(&m, 256, bytes) != 0 || !m) return hipErrorOutOfMemory;      // exactly `bytes`: an overrun meets the sanitizer's red zone
    memset(m, 0xFF, std::min(bytes, (size_t)256 << 20));                  // device memory is not zeroed: NaN as a float, -1 as an int (allocations beyond 256 MB: their head)
    { std::lock_guard<std::mutex> lk(g_mem_mutex); g_device_allocs[m] = bytes; g_device_bytes += bytes; }
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        auto it = g_device_allocs.find(p);
        if (it == g_device_allocs.end()) { fprintf(stderr, "hipcpu: hipFree of a pointer hipMalloc did not return (%p)\n", p); return hipErrorInvalidValue; }
        g_device_bytes -= it->second;
        g_device_allocs.erase(it);
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = bytes ? aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095) : nullptr; return (*p || !bytes) ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s)
{
    submit(s, [dst, src, bytes]() { if (bytes) memmove(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s)
{
    submit(s, [=]() { for (size_t r = 0; r < height; r++) memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width); });
    return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t bytes) { if (bytes) memset(p, v, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s) { submit(s, [p, v, bytes]() { if (bytes) memset(p, v, bytes); }); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new Stream; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { if (s) { delete s->capture; delete s; } return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { return S(s)->capture ? hipErrorStreamCaptureUnsupported : hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st) { *st = S(s)->capture ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode)
{
    Stream* st = S(s);
    if (st->capture || !s) return hipErrorInvalidValue;
    st->capture = new std::vector<std::function<void()>>;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g)
{
    Stream* st = S(s);
    if (!st->capture) { *g = nullptr; return hipErrorInvalidValue; }
    Graph* gr = new Graph;
    gr->ops.swap(*st->capture);
    delete st->capture;
    st->capture = nullptr;
    *g = gr;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, void*, void*, size_t)
{
    if (!g) return hipErrorInvalidValue;
    GraphExec* e = new GraphExec;
    e->ops = g->ops;
    *x = e;
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t x) { delete x; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t s)
{
    if (!x) return hipErrorInvalidValue;
    for (auto& op : x->ops) submit(s, op);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new Event; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new Event; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    if (!e) return hipErrorInvalidValue;
    submit(s, [e]() { e->t = std::chrono::steady_clock::now(); e->recorded = true; });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    if (!a || !b || !a->recorded || !b->recorded) return hipErrorInvalidValue;
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v)
{
    if (a != hipFuncAttributeMaxDynamicSharedMemorySize || v < 0 || (size_t)v > LDS_PER_BLOCK) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(g_attr_mutex);
    g_dyn_lds_limit[f] = (size_t)v;
    return hipSuccess;
}
