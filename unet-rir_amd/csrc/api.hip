// api.hip - the C ABI (include/unetrir.h): TF padding='same' geometry -> tap tables -> launches.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>
#include "kernels.h"

// ---- kernel-selection switches: the only environment variables the library reads, once.  The switches in effect are an IMMUTABLE
//      snapshot behind one atomic pointer: unetrir_set_config publishes a new snapshot (release), every reader takes the pointer
//      (acquire) - a launch being issued on another thread sees the old values or the new ones, never a half-written struct.
//      Replaced snapshots are kept (a few hundred bytes per call of a function only tests and A/B scripts use): a reader may still
//      hold one.
namespace {
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
const unetrir_config* load_config() {
    unetrir_config* c = new unetrir_config;
    c->conv3x3 = env_int("UNETRIR_CONV3X3", 1);
    c->conv3x3g = env_int("UNETRIR_CONV3X3G", 1);
    c->conv3x3g_pair = env_int("UNETRIR_CONV3X3G_PAIR", 1);
    c->conv3x3h = env_int("UNETRIR_CONV3X3H", 1);
    c->conv3x3s = env_int("UNETRIR_CONV3X3S", 1);
    c->conv3x3r = env_int("UNETRIR_CONV3X3R", 1);
    c->stem = env_int("UNETRIR_STEM", 1);
    c->upconv3x3g = env_int("UNETRIR_UPCONV3X3G", 1);
    c->wgrad3x3g = env_int("UNETRIR_WGRAD3X3G", 1);
    c->wgrad3x3r = env_int("UNETRIR_WGRAD3X3R", 1);
    c->wgrad3x3d = env_int("UNETRIR_WGRAD3X3D", 1);
    c->conv3x3d = env_int("UNETRIR_CONV3X3D", 1);
    c->conv3x3p = env_int("UNETRIR_CONV3X3P", 1);
    c->upconv3x3q = env_int("UNETRIR_UPCONV3X3Q", 1);
    c->dyn_tiles = env_int("UNETRIR_DYN_TILES", 1);
    c->head_mfma = env_int("UNETRIR_HEAD_MFMA", 1);
    c->pw1x1 = env_int("UNETRIR_PW1X1", 1);
    c->igemm2 = env_int("UNETRIR_IGEMM2", 1);
    return c;
}
std::atomic<const unetrir_config*>& config_slot() {
    static std::atomic<const unetrir_config*> p{load_config()};
    return p;
}
}  // namespace
const unetrir_config& unetrir_cfg() { return *config_slot().load(std::memory_order_acquire); }
extern "C" int unetrir_get_config(unetrir_config* out) {
    if (!out) return UNETRIR_EINVAL;
    *out = unetrir_cfg();
    return 0;
}
extern "C" int unetrir_set_config(const unetrir_config* in) {
    if (!in) return UNETRIR_EINVAL;
    config_slot().store(new unetrir_config(*in), std::memory_order_release);
    return 0;
}

#ifdef UNETRIR_ABLATIONS
int g_unetrir_abl = 0;
extern "C" int unetrir_abl_set(int v) { g_unetrir_abl = v; return 0; }
// A stand-in for a communication kernel beside the step (ablation build only): n workgroups that hold `lds` bytes of LDS and
// spin for `cycles` clock cycles - CUs on which a 158 KB persistent workgroup cannot be placed meanwhile.
__global__ void abl_hog_kernel(long long cycles, int lds, unsigned* sink) {
    extern __shared__ unsigned hog_smem[];
    if (lds > 0) hog_smem[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < cycles) acc += hog_smem[(threadIdx.x + acc) & 63];
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
extern "C" int unetrir_abl_hog(int n, long long cycles, int lds, void* sink, void* stream) {
    hipLaunchKernelGGL(abl_hog_kernel, dim3(n), dim3(256), (size_t)lds, (hipStream_t)stream, cycles, lds, (unsigned*)sink);
    return (int)hipGetLastError();
}
#endif

// ---- per-stream ticket slots (kernels.h): a static device array, one slot per (device, stream) in use: the tile tickets of the
//      persistent convolution kernels (64 group counters + 1 count of finished workgroups).  Zero between launches: every kernel
//      that uses a slot leaves it cleared.
__device__ unsigned g_sched_slots[128][80];
// A __device__ symbol has one instance PER DEVICE: the table below is keyed by the device that is current at the launch (the
// reference's own process shape is one process driving several GPUs, main_training.py:56), and a slot by (device, stream).
// More than 16 devices or 128 streams in use on one device: no slot (-1) - the callers then take their slot-free path.
// The table is append-only, so the lookup every persistent-kernel launch performs takes NO lock: `used` is published with release
// after the owner entry is written; only a miss (a stream's first launch, a device's first use) takes the mutex.
namespace {
struct SlotTable { std::atomic<unsigned*> sched{nullptr}; hipStream_t owner[128]; std::atomic<int> used{0}; };
constexpr int MAX_DEV = 16;
std::mutex g_slot_mu;
SlotTable g_slot_tab[MAX_DEV];

// index of the slot of (current device, s), or -1; *base receives the device's slot array
int slot_index(hipStream_t s, unsigned** base) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;
    SlotTable& d = g_slot_tab[dev];
    unsigned* sched = d.sched.load(std::memory_order_acquire);
    if (sched) {
        const int n = d.used.load(std::memory_order_acquire);
        for (int i = 0; i < n; ++i) if (d.owner[i] == s) { *base = sched; return i; }
    }
    std::lock_guard<std::mutex> lk(g_slot_mu);           // miss: first launch of this stream (or first use of this device)
    sched = d.sched.load(std::memory_order_relaxed);
    if (!sched) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sched_slots)) != hipSuccess) return -1;              // resolve on the current device
        // zeroed once, synchronously, on first use of this device (a stream that is being captured into a HIP graph refuses the
        // call: no slot then, and nothing is cached - the engines call unetrir_reset_tile_tickets() when they are built)
        if (hipMemset(p, 0, sizeof(unsigned) * 128 * 80) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        sched = (unsigned*)p;
        d.sched.store(sched, std::memory_order_release);
    }
    *base = sched;
    const int n = d.used.load(std::memory_order_relaxed);
    for (int i = 0; i < n; ++i) if (d.owner[i] == s) return i;      // another thread appended it meanwhile
    if (n == 128) return -1;
    d.owner[n] = s;
    d.used.store(n + 1, std::memory_order_release);
    return n;
}
}  // namespace

unsigned* sched_slot(hipStream_t s) {
    if (!unetrir_cfg().dyn_tiles) return nullptr;
    unsigned* base = nullptr;
    const int i = slot_index(s, &base);
    return i < 0 ? nullptr : base + i * 80;
}

// Host-side reset of every ticket slot of the current device (after toggling dyn_tiles, or after a launch failed): call with the
// device idle.  The kernels clear their own slot at the end of every launch, so a healthy run never needs it.
extern "C" int unetrir_reset_tile_tickets(void) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sched_slots)) != hipSuccess) return UNETRIR_EINVAL;
    return (int)hipMemset(p, 0, sizeof(unsigned) * 128 * 80);
}

namespace {

struct Same { int out, before; };
// tf.nn.convolution padding='same': out = ceil(in/s), pad_total = max((out-1)*s + k - in, 0),
// pad_before = pad_total / 2 (dl_models/u_net.py:269-276 relies on it for the strided convs).
inline Same same_geom(int n_in, int k, int s) {
    Same r;
    r.out = (n_in + s - 1) / s;
    int total = (r.out - 1) * s + k - n_in;
    if (total < 0) total = 0;
    r.before = total / 2;
    return r;
}

inline uint32_t pack_tap(int dy, int dx, int widx) {
    return (uint32_t)(uint8_t)(int8_t)dy | ((uint32_t)(uint8_t)(int8_t)dx << 8) | ((uint32_t)widx << 16);
}

inline bool geom_ok(const unetrir_conv_geom* g) {
    return g && g->B > 0 && g->H > 0 && g->W > 0 && g->Cin > 0 && g->Cout > 0 && g->k >= 1 && g->k <= 6 &&
           (g->stride == 1 || g->stride == 2);
}

// ---- profiling (the only process-global state in the library; off by default) ----
struct ProfRec { int fam, tag; hipEvent_t e0, e1; double flops; };
std::mutex g_prof_mu;
bool g_prof_on = false;
unsigned g_prof_mask = ~0u;       // families that get brackets (unetrir_prof_enable(2): forward convolutions only)
std::vector<ProfRec> g_prof;

struct ProfScope {
    bool on; ProfRec r; hipStream_t s;
    // tag: a second family the bracket is ALSO counted under (UNETRIR_FAM_DOMINANT: launches served by the dominant kernel), or -1
    ProfScope(int fam, double flops, hipStream_t st, int tag = -1) : on(g_prof_on && ((g_prof_mask >> fam) & 1u)), s(st) {
        if (!on) return;
        r.fam = fam; r.tag = tag; r.flops = flops;
        hipEventCreate(&r.e0); hipEventCreate(&r.e1);
        hipEventRecord(r.e0, s);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(r.e1, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};

inline double conv_flops(const unetrir_conv_geom* g) {
    const Same sy = same_geom(g->H, g->k, g->stride), sx = same_geom(g->W, g->k, g->stride);
    return 2.0 * g->B * sy.out * sx.out * (double)g->Cout * g->Cin * g->k * g->k;
}
inline int conv_family(const unetrir_conv_geom* g, int fam) {
    return (g->Cin <= 8 || g->Cout <= 8) ? 5 : fam;     // zero-padded stem / head: not part of the MFMA roofline figure
}

// element-type policies: fp32 and bf16-storage variants share the tap-table construction
// 3x3 stride-1 layers go to the patch-staged kernel (conv3x3.hip) when its 8 x 32 pixel tiles cover the image well
// (config switch conv3x3 = 0 forces the tap-table kernels, for A/B measurements)
inline bool use_conv3x3(int k, int stride, int H, int W) {
    if (!unetrir_cfg().conv3x3 || k != 3 || stride != 1) return false;
    const double util = (double)H * W / ((double)((H + 7) / 8 * 8) * ((W + 31) / 32 * 32));
    return util >= 0.7;
}

// 1x1 layers in bf16 storage: the register-streaming kernel (pw1x1.hip).  Forward form: gather at the convolution's stride;
// data-gradient form (also Conv2DTranspose forward): scatter at the stride, the other pixels of a 2 x 2 cell filled with bias
// (+ addend) - for k = 1 TF 'same' has no padding at either stride, so input pixel = stride * output pixel exactly.
inline PwArgs pw_fwd_args(const unetrir_conv_geom* g, const void* x, int ldx, const void* w, const float* bias, const void* addend,
                          int ldadd, void* y, int ldy, float* colstat) {
    PwArgs a{};
    const Same sy = same_geom(g->H, 1, g->stride), sx = same_geom(g->W, 1, g->stride);
    a.in = (const __bf16*)x; a.ldi = ldx; a.IH = g->H; a.IW = g->W;
    a.w = (const __bf16*)w; a.bias = bias; a.addend = (const __bf16*)addend; a.ldadd = ldadd;
    a.out = (__bf16*)y; a.ldo = ldy; a.OH = sy.out; a.OW = sx.out;
    a.B = g->B; a.PH = sy.out; a.PW = sx.out; a.SI = g->stride; a.SO = 1; a.fill = 0;
    a.C = g->Cin; a.N = g->Cout; a.colstat = colstat;
    return a;
}
inline PwArgs pw_dgrad_args(const unetrir_conv_geom* g, const void* dy, int lddy, const void* wt, const float* bias, const void* addend,
                            int ldadd, void* dx, int lddx, float* colstat) {
    PwArgs a{};
    const Same sy = same_geom(g->H, 1, g->stride), sx = same_geom(g->W, 1, g->stride);
    a.in = (const __bf16*)dy; a.ldi = lddy; a.IH = sy.out; a.IW = sx.out;
    a.w = (const __bf16*)wt; a.bias = bias; a.addend = (const __bf16*)addend; a.ldadd = ldadd;
    a.out = (__bf16*)dx; a.ldo = lddx; a.OH = g->H; a.OW = g->W;
    a.B = g->B; a.PH = sy.out; a.PW = sx.out; a.SI = 1; a.SO = g->stride; a.fill = g->stride == 2 ? 1 : 0;
    a.C = g->Cout; a.N = g->Cin; a.colstat = colstat;
    return a;
}

struct F32 {
    static constexpr int is_bf16 = 0;
    using T = float; using Args = IgemmArgs;
    static int launch(const Args& a, hipStream_t s) { return launch_igemm_fwd(a, s); }
    static int launch_classes(const Args* a, hipStream_t s) {
        for (int i = 0; i < 4; ++i) { const int err = launch_igemm_fwd(a[i], s); if (err) return err; }
        return 0;
    }
};
struct BF16 {
    static constexpr int is_bf16 = 1;
    using T = __bf16; using Args = IgemmArgsH;
    static int launch(const Args& a, hipStream_t s) { return launch_igemm_fwd_bf16(a, s); }
    static int launch_classes(const Args* a, hipStream_t s) {
        return launch_igemm_fwd_bf16_x4(a, s);          // the four parity classes share one grid
    }
};

// ---- Conv2D forward: iteration grid = output grid ----
template <class P>
int conv_fwd_impl(const unetrir_conv_geom* g, const typename P::T* x, int ldx, const typename P::T* w, const float* bias,
                  const typename P::T* addend, int ldadd, typename P::T* y, int ldy, hipStream_t s, float* colstat = nullptr,
                  const void* wpk = nullptr) {
    if (g->k == 3 && g->stride == 1) {
        Conv3Args c{};
        c.colstat = colstat;
        c.in = x; c.ldi = ldx; c.w = w; c.bias = bias; c.addend = addend; c.ldadd = ldadd; c.out = y; c.ldo = ldy;
        c.B = g->B; c.H = g->H; c.W = g->W; c.C = g->Cin; c.N = g->Cout;
        c.flip = UNETRIR_ABL(UNETRIR_ABL_HOST(), 256) ? 2 : 0;      // ablation build only: register-staged kernel without its stores
        // narrow images (the 16 x 16 level) would half-fill the 32-column tiles: bf16 has a paired-image tile for them
        if (use_conv3x3(g->k, g->stride, g->H, g->W) || (P::is_bf16 && conv3x3g_pair_applies(c))) return launch_conv3x3(c, P::is_bf16, s);
    }
    if constexpr (P::is_bf16) {
        if (g->k == 3 && g->stride == 2 && !colstat) {      // strided Conv2D / Conv2DTranspose data gradient: persistent LDS-DMA kernel
            Conv3Args c{};
            c.in = x; c.ldi = ldx; c.w = w; c.bias = bias; c.addend = addend; c.ldadd = ldadd; c.out = y; c.ldo = ldy;
            c.B = g->B; c.H = g->H; c.W = g->W; c.C = g->Cin; c.N = g->Cout;
            c.wpk = (g->Cin % 64 == 0 && g->Cout % 64 == 0) ? wpk : nullptr;
            if (conv3x3d_applies(c)) return launch_conv3x3d_bf16(c, s);
        }
    }
    if constexpr (P::is_bf16) {
        if (g->k == 1) {
            const PwArgs pw = pw_fwd_args(g, x, ldx, w, bias, addend, ldadd, y, ldy, colstat);
            if (pw1x1_applies(pw)) return launch_pw1x1_bf16(pw, s);
        }
    }
    const Same sy = same_geom(g->H, g->k, g->stride), sx = same_geom(g->W, g->k, g->stride);
    typename P::Args a{};
    a.g.B = g->B; a.g.PH = sy.out; a.g.PW = sx.out;
    a.g.IH = g->H; a.g.IW = g->W; a.g.C = g->Cin; a.g.ldi = ldx;
    a.g.OH = sy.out; a.g.OW = sx.out; a.g.N = g->Cout; a.g.ldo = ldy;
    a.g.SI = g->stride; a.g.SO = 1; a.g.ooy = 0; a.g.oox = 0;
    a.g.ntaps = g->k * g->k; a.g.wtaps = g->k * g->k;
    for (int kh = 0; kh < g->k; ++kh)
        for (int kw = 0; kw < g->k; ++kw)
            a.g.tap[kh * g->k + kw] = pack_tap(kh - sy.before, kw - sx.before, kh * g->k + kw);
    a.in = x; a.w = w; a.bias = bias; a.addend = addend; a.ldadd = ldadd; a.out = y;
    if constexpr (P::is_bf16) a.colstat = colstat;      // tap-table kernel: one row of column statistics per 128-pixel tile
    return P::launch(a, s);
}

// ---- Conv2D data gradient (also Conv2DTranspose forward when `bias` is given) ----
// dx[q][ci] = sum_t sum_co dy[p][co] * wt[ci][t][co]  with q = p*s + (k_t - pad_before)
template <class P>
int conv_dgrad_impl(const unetrir_conv_geom* g, const typename P::T* dy, int lddy, const typename P::T* wt, const float* bias,
                    const typename P::T* addend, int ldadd, typename P::T* dx, int lddx, hipStream_t s, float* colstat = nullptr) {
    if (g->k == 3 && g->stride == 1) {     // dgrad of a stride-1 3x3 conv = the same conv with flipped taps
        Conv3Args c{};
        c.colstat = colstat;
        c.in = dy; c.ldi = lddy; c.w = wt; c.bias = bias; c.addend = addend; c.ldadd = ldadd; c.out = dx; c.ldo = lddx;
        c.B = g->B; c.H = g->H; c.W = g->W; c.C = g->Cout; c.N = g->Cin; c.flip = 1;
        if (use_conv3x3(g->k, g->stride, g->H, g->W) || (P::is_bf16 && conv3x3g_pair_applies(c))) return launch_conv3x3(c, P::is_bf16, s);
    }
    if constexpr (P::is_bf16) {
        if (g->k == 1) {
            const PwArgs pw = pw_dgrad_args(g, dy, lddy, wt, bias, addend, ldadd, dx, lddx, colstat);
            if (pw1x1_applies(pw)) return launch_pw1x1_bf16(pw, s);
        }
    }
    const Same sy = same_geom(g->H, g->k, g->stride), sx = same_geom(g->W, g->k, g->stride);
    typename P::Args a{};
    a.g.B = g->B;
    a.g.IH = sy.out; a.g.IW = sx.out; a.g.C = g->Cout; a.g.ldi = lddy;
    a.g.OH = g->H; a.g.OW = g->W; a.g.N = g->Cin; a.g.ldo = lddx;
    a.g.wtaps = g->k * g->k;
    a.in = dy; a.w = wt; a.bias = bias; a.addend = addend; a.ldadd = ldadd; a.out = dx;
    if constexpr (P::is_bf16) a.colstat = colstat;
    if (g->stride == 1) {
        a.g.PH = g->H; a.g.PW = g->W; a.g.SI = 1; a.g.SO = 1; a.g.ooy = 0; a.g.oox = 0;
        a.g.ntaps = g->k * g->k;
        for (int kh = 0; kh < g->k; ++kh)
            for (int kw = 0; kw < g->k; ++kw)
                a.g.tap[kh * g->k + kw] = pack_tap(-(kh - sy.before), -(kw - sx.before), kh * g->k + kw);
        return P::launch(a, s);
    }
    // stride 2, 3x3, even sizes (pad_before 0): all four output parity classes in one patch-staged launch (upconv3x3.hip)
    if (g->k == 3 && sy.before == 0 && sx.before == 0 && g->H == 2 * sy.out && g->W == 2 * sx.out) {
        Conv3Args c{};
        c.in = dy; c.ldi = lddy; c.w = wt; c.bias = bias; c.addend = addend; c.ldadd = ldadd; c.out = dx; c.ldo = lddx;
        c.B = g->B; c.H = sy.out; c.W = sx.out; c.C = g->Cout; c.N = g->Cin; c.flip = 0;
        // half-empty tiles (16-wide coarse grids) stay on the four tap-table launches: measured 0.2 ms/step faster than this kernel there
        const double min_util = 0.7;
        const double util = (double)c.H * c.W / ((double)((c.H + 7) / 8 * 8) * ((c.W + 31) / 32 * 32));
        if (!colstat && (use_conv3x3(3, 1, sy.out, sx.out) || (P::is_bf16 && upconv3x3g_applies(c) && util >= min_util)))
            return launch_upconv3x3(c, P::is_bf16, s);
    }
    // stride 2: one launch per output parity class (ay, ax); q = 2p' + a, p = p' + (a - off)/2
    a.g.PH = (g->H + 1) / 2; a.g.PW = (g->W + 1) / 2; a.g.SI = 1; a.g.SO = 2;
    typename P::Args cls[4];
    for (int ay = 0; ay < 2; ++ay)
        for (int ax = 0; ax < 2; ++ax) {
            int nt = 0;
            for (int kh = 0; kh < g->k; ++kh) {
                const int offy = kh - sy.before;
                if (((offy - ay) & 1) != 0) continue;
                for (int kw = 0; kw < g->k; ++kw) {
                    const int offx = kw - sx.before;
                    if (((offx - ax) & 1) != 0) continue;
                    a.g.tap[nt++] = pack_tap((ay - offy) / 2, (ax - offx) / 2, kh * g->k + kw);
                }
            }
            a.g.ntaps = nt; a.g.ooy = ay; a.g.oox = ax;
            if constexpr (P::is_bf16) {      // column statistics: the four classes write consecutive row ranges
                if (colstat) a.colstat = colstat + (size_t)(ay * 2 + ax) * igemm_colstat_rows((long long)g->B * a.g.PH * a.g.PW, g->Cin, 4) * g->Cin * 2;
            }
            cls[ay * 2 + ax] = a;
        }
    return P::launch_classes(cls, s);       // bf16: the four classes share one grid
}

void wgrad_args(const unetrir_conv_geom* g, int ldx, int lddy, WgradArgs* a) {
    const Same sy = same_geom(g->H, g->k, g->stride), sx = same_geom(g->W, g->k, g->stride);
    a->g.B = g->B; a->g.PH = sy.out; a->g.PW = sx.out;
    a->g.IH = g->H; a->g.IW = g->W; a->g.C = g->Cin; a->g.ldi = ldx;
    a->g.N = g->Cout; a->g.SI = g->stride;
    a->g.ntaps = g->k * g->k; a->g.wtaps = g->k * g->k;
    for (int kh = 0; kh < g->k; ++kh)
        for (int kw = 0; kw < g->k; ++kw)
            a->g.tap[kh * g->k + kw] = pack_tap(kh - sy.before, kw - sx.before, kh * g->k + kw);
    a->lddy = lddy;
}

size_t wgrad_ws_bytes(const unetrir_conv_geom* g) {
    if (g->k == 3) {
        const Same sy = same_geom(g->H, 3, g->stride), sx = same_geom(g->W, 3, g->stride);
        return wgrad3x3_ws_bytes(g->stride, g->B, sy.out, sx.out, g->Cout, g->Cin);
    }
    WgradArgs a{};
    wgrad_args(g, g->Cin, g->Cout, &a);
    int ns; long long per;
    wgrad_plan(a.g, &ns, &per);
    size_t bytes = (size_t)ns * g->Cout * g->k * g->k * g->Cin * sizeof(float);
    if (g->k == 1) {          // the bf16 1x1 kernel has its own split plan: the workspace serves both storage modes
        const Same sy = same_geom(g->H, 1, g->stride), sx = same_geom(g->W, 1, g->stride);
        const size_t b16 = wgrad1x1_bf16_ws_bytes(g->B, sy.out, sx.out, g->Cout, g->Cin);
        if (b16 > bytes) bytes = b16;
    }
    return bytes;
}

int conv_wgrad_impl(const unetrir_conv_geom* g, const float* x, int ldx, const float* dy, int lddy, float* dw,
                    float reg, const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    if (g->k == 3) {   // 3x3: halo-staged patch kernel (wgrad3x3.hip)
        const Same sy = same_geom(g->H, 3, g->stride), sx = same_geom(g->W, 3, g->stride);
        Wgrad3Args a3{};
        a3.x = x; a3.ldx = ldx; a3.IH = g->H; a3.IW = g->W;
        a3.dy = dy; a3.lddy = lddy; a3.OH = sy.out; a3.OW = sx.out;
        a3.B = g->B; a3.C = g->Cin; a3.N = g->Cout;
        a3.pad_t = sy.before; a3.pad_l = sx.before;
        return launch_wgrad3x3(a3, g->stride, dw, reg, w, ws, ws_bytes, s);
    }
    WgradArgs a{};
    wgrad_args(g, ldx, lddy, &a);
    a.x = x; a.dy = dy;
    return launch_igemm_wgrad(a, dw, reg, w, ws, ws_bytes, s);
}

int conv_wgrad_bf16_impl(const unetrir_conv_geom* g, const __bf16* x, int ldx, const __bf16* dy, int lddy, float* dw, float reg,
                         const float* w, void* ws, size_t ws_bytes, hipStream_t s) {
    if (g->k != 3 && g->k != 1) {      // other kernel sizes (kernels = 6: the reference's constructor default): the tap-table weight
                                       // gradient on the bf16 tensors as stored, fp32 MFMA arithmetic
        WgradArgs a{};
        wgrad_args(g, ldx, lddy, &a);
        a.x = (const float*)x; a.dy = (const float*)dy;
        return launch_igemm_wgrad(a, dw, reg, w, ws, ws_bytes, s, 1);
    }
    const Same sy = same_geom(g->H, g->k, g->stride), sx = same_geom(g->W, g->k, g->stride);
    Wgrad3ArgsH a3{};
    a3.x = x; a3.ldx = ldx; a3.IH = g->H; a3.IW = g->W;
    a3.dy = dy; a3.lddy = lddy; a3.OH = sy.out; a3.OW = sx.out;
    a3.B = g->B; a3.C = g->Cin; a3.N = g->Cout;
    a3.pad_t = sy.before; a3.pad_l = sx.before;
    if (g->k == 1) return launch_wgrad1x1_bf16(a3, g->stride, dw, reg, w, ws, ws_bytes, s);
    return launch_wgrad3x3_bf16(a3, g->stride, dw, reg, w, ws, ws_bytes, s);
}

// Conv2DTranspose(k, s=2, 'same') on an H x W input is the adjoint of Conv2D(k, s=2, 'same') that maps the
// 2H x 2W grid back to H x W: swap the channel roles and double the spatial size.
inline unetrir_conv_geom adjoint_geom(const unetrir_conv_geom* g) {
    unetrir_conv_geom c = *g;
    c.H = g->H * g->stride; c.W = g->W * g->stride;
    c.Cin = g->Cout; c.Cout = g->Cin;
    return c;
}

inline bool ld_ok(int ld, int c) { return ld >= c && (ld & 3) == 0; }
inline bool ldh_ok(int ld, int c) { return ld >= c && (ld & 7) == 0; }     // bf16: 16-byte rows

}  // namespace

extern "C" {

int unetrir_abi_version(void) { return UNETRIR_ABI_VERSION; }

int unetrir_conv2d_fwd_f32(const unetrir_conv_geom* g, const float* x, int ldx, const float* w, const float* bias,
                           const float* addend, int ldadd, float* y, int ldy, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !w || !y || (g->Cin & 3) || !ld_ok(ldx, g->Cin) || ldy < g->Cout) return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(g), (hipStream_t)stream);
    return conv_fwd_impl<F32>(g, x, ldx, w, bias, addend, ldadd, y, ldy, (hipStream_t)stream);
}

int unetrir_conv2d_dgrad_f32(const unetrir_conv_geom* g, const float* dy, int lddy, const float* wt,
                             const float* addend, int ldadd, float* dx, int lddx, unetrir_stream_t stream) {
    if (!geom_ok(g) || !dy || !wt || !dx || (g->Cout & 3) || !ld_ok(lddy, g->Cout) || lddx < g->Cin) return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_DGRAD), conv_flops(g), (hipStream_t)stream);
    return conv_dgrad_impl<F32>(g, dy, lddy, wt, nullptr, addend, ldadd, dx, lddx, (hipStream_t)stream);
}

size_t unetrir_conv2d_wgrad_ws_bytes(const unetrir_conv_geom* g) { return geom_ok(g) ? wgrad_ws_bytes(g) : 0; }

int unetrir_conv2d_wgrad_f32(const unetrir_conv_geom* g, const float* x, int ldx, const float* dy, int lddy, float* dw,
                             float reg_coef, const float* w, void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !dy || !dw || (g->Cin & 3) || !ld_ok(ldx, g->Cin) || (lddy & 3) || lddy < g->Cout ||
        (reg_coef != 0.f && !w))
        return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_WGRAD), conv_flops(g), (hipStream_t)stream);
    return conv_wgrad_impl(g, x, ldx, dy, lddy, dw, reg_coef, w, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_conv2d_transpose_fwd_f32(const unetrir_conv_geom* g, const float* x, int ldx, const float* wt,
                                     const float* bias, float* y, int ldy, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !wt || !y || (g->Cin & 3) || !ld_ok(ldx, g->Cin) || ldy < g->Cout)
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(&c), (hipStream_t)stream);
    return conv_dgrad_impl<F32>(&c, x, ldx, wt, bias, nullptr, 0, y, ldy, (hipStream_t)stream);
}

int unetrir_conv2d_transpose_dgrad_f32(const unetrir_conv_geom* g, const float* dy, int lddy, const float* w,
                                       const float* addend, int ldadd, float* dx, int lddx, unetrir_stream_t stream) {
    if (!geom_ok(g) || !dy || !w || !dx || (g->Cout & 3) || !ld_ok(lddy, g->Cout) || lddx < g->Cin)
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_DGRAD), conv_flops(&c), (hipStream_t)stream);
    return conv_fwd_impl<F32>(&c, dy, lddy, w, nullptr, addend, ldadd, dx, lddx, (hipStream_t)stream);
}

size_t unetrir_conv2d_transpose_wgrad_ws_bytes(const unetrir_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    const unetrir_conv_geom c = adjoint_geom(g);
    return wgrad_ws_bytes(&c);
}

int unetrir_conv2d_transpose_wgrad_f32(const unetrir_conv_geom* g, const float* x, int ldx, const float* dy, int lddy,
                                       float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes,
                                       unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !dy || !dw || (g->Cout & 3) || !ld_ok(lddy, g->Cout) || (ldx & 3) ||
        ldx < g->Cin || (reg_coef != 0.f && !w))
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_WGRAD), conv_flops(&c), (hipStream_t)stream);
    // adjoint conv: its "x" is our dy (2H x 2W, Cout channels), its "dy" is our x (H x W, Cin channels)
    return conv_wgrad_impl(&c, dy, lddy, x, ldx, dw, reg_coef, w, ws, ws_bytes, (hipStream_t)stream);
}

/* ---- bf16-storage variants: x / dy / y / dx and the weight work copies are bf16, bias fp32, weight gradients fp32 ---- */
int unetrir_conv3x3_kernel_id_bf16(const unetrir_conv_geom* g, int dgrad, int ld_in);
// profiling only: launches the dominant kernel of the step (conv3x3p) serves are also counted under UNETRIR_FAM_DOMINANT
static inline int dominant_tag(const unetrir_conv_geom* g, int dgrad, int ld_in) {
    if (!g_prof_on || !geom_ok(g) || g->k != 3 || g->stride != 1) return -1;
    return unetrir_conv3x3_kernel_id_bf16(g, dgrad, ld_in) == UNETRIR_K3_CONV3X3P ? UNETRIR_FAM_DOMINANT : -1;
}
int unetrir_conv2d_fwd_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* w, const float* bias,
                            const unetrir_bf16* addend, int ldadd, unetrir_bf16* y, int ldy, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !w || !y || (g->Cin & 7) || !ldh_ok(ldx, g->Cin) || ldy < g->Cout) return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(g), (hipStream_t)stream, dominant_tag(g, 0, ldx));
    return conv_fwd_impl<BF16>(g, (const __bf16*)x, ldx, (const __bf16*)w, bias, (const __bf16*)addend, ldadd, (__bf16*)y, ldy,
                               (hipStream_t)stream);
}

size_t unetrir_conv3x3s2_packed_elems(int N, int C) {
    if (N <= 0 || C <= 0 || (N & 63) || (C & 63)) return 0;
    return (size_t)((N + 127) / 128) * 128 * 9 * C;
}

int unetrir_conv2d_fwd_packed_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* w,
                                   const unetrir_bf16* w_packed, const float* bias, const unetrir_bf16* addend, int ldadd,
                                   unetrir_bf16* y, int ldy, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !w || !y || (g->Cin & 7) || !ldh_ok(ldx, g->Cin) || ldy < g->Cout) return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(g), (hipStream_t)stream);
    return conv_fwd_impl<BF16>(g, (const __bf16*)x, ldx, (const __bf16*)w, bias, (const __bf16*)addend, ldadd, (__bf16*)y, ldy,
                               (hipStream_t)stream, nullptr, w_packed);
}

int unetrir_conv2d_dgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy, const unetrir_bf16* wt,
                              const unetrir_bf16* addend, int ldadd, unetrir_bf16* dx, int lddx, unetrir_stream_t stream) {
    if (!geom_ok(g) || !dy || !wt || !dx || (g->Cout & 7) || !ldh_ok(lddy, g->Cout) || lddx < g->Cin) return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_DGRAD), conv_flops(g), (hipStream_t)stream, dominant_tag(g, 1, lddy));
    return conv_dgrad_impl<BF16>(g, (const __bf16*)dy, lddy, (const __bf16*)wt, nullptr, (const __bf16*)addend, ldadd, (__bf16*)dx,
                                 lddx, (hipStream_t)stream);
}

/* Fused column statistics: the 3x3 stride-1 kernels conv3x3g / conv3x3r<4,1> can emit, per 16 x 32 pixel tile, the
 * per-channel (sum, sum of squares) of the bf16 output they store.  rows == 0: the kernel serving this layer cannot. */
static long long colstat_rows(const unetrir_conv_geom* g, int dgrad, int ld_in) {
    if (!geom_ok(g)) return 0;
    const Same sy = same_geom(g->H, g->k, g->stride), sx = same_geom(g->W, g->k, g->stride);
    // rows the tap-table kernel writes: one per 128-pixel tile of its iteration grid (forward: the output grid; data gradient at
    // stride 1: the input grid).  Strided data gradients have no fused statistics through this entry point.
    const long long igemm_rows = dgrad ? (g->stride == 1 ? igemm_colstat_rows((long long)g->B * g->H * g->W, g->Cin) : 0)
                                       : igemm_colstat_rows((long long)g->B * sy.out * sx.out, g->Cout);
    if (g->k == 1) {          // the register-streaming 1x1 kernel: one row per persistent workgroup
        const PwArgs pw = dgrad ? pw_dgrad_args(g, nullptr, ld_in, nullptr, nullptr, nullptr, 0, nullptr, g->Cin, nullptr)
                                : pw_fwd_args(g, nullptr, ld_in, nullptr, nullptr, nullptr, 0, nullptr, g->Cout, nullptr);
        if (pw1x1_applies(pw)) return pw1x1_colstat_rows(pw);
    }
    if (g->k != 3 || g->stride != 1) return (g->k == 3 && g->stride == 2 && !dgrad) ? 0 : igemm_rows;    // 3x3 stride 2 forward: conv3x3d (none)
    Conv3Args c{};
    c.B = g->B; c.H = g->H; c.W = g->W; c.ldi = ld_in;
    c.C = dgrad ? g->Cout : g->Cin; c.N = dgrad ? g->Cin : g->Cout; c.flip = dgrad ? 1 : 0;
    if (conv3x3g_pair_applies(c)) return conv3x3g_colstat_rows(c);          // two images per tile row
    if (!use_conv3x3(g->k, g->stride, g->H, g->W)) return igemm_rows;
    if (!conv3x3_has_colstat(c)) return 0;
    if (conv3x3s_applies(c)) return conv3x3s_colstat_rows(c);               // one row per persistent workgroup
    if (unetrir_cfg().conv3x3g && conv3x3p_applies(c)) return conv3x3p_colstat_rows(c);    // one row per group of N / 128 workgroups
    return (long long)g->B * ((g->H + 15) / 16) * ((g->W + 31) / 32);
}
// Conv2DTranspose forward (g: the transposed layer's own geometry).  Stride 1: the data-gradient form of the adjoint conv on
// the same grid.  Stride 2: the four output-parity classes of the tap-table path, each with its own row range; the patch-staged
// 3x3 kernels of that path (upconv3x3*) have no fused statistics, so a 3x3 stride-2 layer on even sizes reports 0 unless the
// caller forces the tap-table path by asking for statistics (the launch then takes it).
static long long colstat_rows_transpose(const unetrir_conv_geom* g, int ld_in) {
    if (!geom_ok(g)) return 0;
    if (g->stride == 1) {
        unetrir_conv_geom c = *g;
        c.Cin = g->Cout; c.Cout = g->Cin;
        return colstat_rows(&c, 1, ld_in);
    }
    if (g->k == 3) return 0;
    if (g->k == 1) {
        const unetrir_conv_geom c = adjoint_geom(g);
        const PwArgs pw = pw_dgrad_args(&c, nullptr, ld_in, nullptr, nullptr, nullptr, 0, nullptr, g->Cout, nullptr);
        if (pw1x1_applies(pw)) return pw1x1_colstat_rows(pw);
    }
    return 4 * igemm_colstat_rows((long long)g->B * g->H * g->W, g->Cout, 4);
}
long long unetrir_conv2d_colstat_rows_bf16(const unetrir_conv_geom* g, int dgrad, int ld_in) { return colstat_rows(g, dgrad, ld_in); }

int unetrir_conv3x3_kernel_id_bf16(const unetrir_conv_geom* g, int dgrad, int ld_in) {
    if (!geom_ok(g) || g->k != 3 || g->stride != 1) return UNETRIR_K3_TAPTABLE;
    Conv3Args c{};
    c.B = g->B; c.H = g->H; c.W = g->W; c.ldi = ld_in;
    c.C = dgrad ? g->Cout : g->Cin; c.N = dgrad ? g->Cin : g->Cout; c.flip = dgrad ? 1 : 0;
    const bool dma = unetrir_cfg().conv3x3g != 0;
    if (conv3x3g_pair_applies(c) && dma) return UNETRIR_K3_CONV3X3G_PAIR;
    if (!use_conv3x3(g->k, g->stride, g->H, g->W)) return UNETRIR_K3_TAPTABLE;
    if (unetrir_cfg().stem && stem3x3_applies(c)) return UNETRIR_K3_STEM;
    if (dma && conv3x3p_applies(c)) return UNETRIR_K3_CONV3X3P;
    if (dma && conv3x3g_applies(c)) return UNETRIR_K3_CONV3X3G;
    if (conv3x3s_applies(c)) return UNETRIR_K3_CONV3X3S;
    if (conv3x3h_applies(c)) return UNETRIR_K3_CONV3X3H;
    return UNETRIR_K3_CONV3X3R;
}

int unetrir_conv2d_fwd_colstat_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* w, const float* bias,
                                    const unetrir_bf16* addend, int ldadd, unetrir_bf16* y, int ldy, float* colstat,
                                    unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !w || !y || !colstat || (g->Cin & 7) || !ldh_ok(ldx, g->Cin) || ldy < g->Cout || colstat_rows(g, 0, ldx) == 0)
        return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(g), (hipStream_t)stream, dominant_tag(g, 0, ldx));
    return conv_fwd_impl<BF16>(g, (const __bf16*)x, ldx, (const __bf16*)w, bias, (const __bf16*)addend, ldadd, (__bf16*)y, ldy,
                               (hipStream_t)stream, colstat);
}

int unetrir_conv2d_dgrad_colstat_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy, const unetrir_bf16* wt,
                                      const unetrir_bf16* addend, int ldadd, unetrir_bf16* dx, int lddx, float* colstat,
                                      unetrir_stream_t stream) {
    if (!geom_ok(g) || !dy || !wt || !dx || !colstat || (g->Cout & 7) || !ldh_ok(lddy, g->Cout) || lddx < g->Cin ||
        colstat_rows(g, 1, lddy) == 0)
        return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_DGRAD), conv_flops(g), (hipStream_t)stream, dominant_tag(g, 1, lddy));
    return conv_dgrad_impl<BF16>(g, (const __bf16*)dy, lddy, (const __bf16*)wt, nullptr, (const __bf16*)addend, ldadd, (__bf16*)dx,
                                 lddx, (hipStream_t)stream, colstat);
}

long long unetrir_conv2d_transpose_colstat_rows_bf16(const unetrir_conv_geom* g, int ld_in) { return colstat_rows_transpose(g, ld_in); }

int unetrir_conv2d_transpose_fwd_colstat_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* wt,
                                              const float* bias, unetrir_bf16* y, int ldy, float* colstat, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !wt || !y || !colstat || (g->Cin & 7) || !ldh_ok(ldx, g->Cin) || ldy < g->Cout || colstat_rows_transpose(g, ldx) == 0)
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(&c), (hipStream_t)stream);
    return conv_dgrad_impl<BF16>(&c, (const __bf16*)x, ldx, (const __bf16*)wt, bias, nullptr, 0, (__bf16*)y, ldy, (hipStream_t)stream, colstat);
}

int unetrir_conv2d_wgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* dy, int lddy,
                              float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !dy || !dw || (g->Cin & 7) || (g->Cout & 7) || !ldh_ok(ldx, g->Cin) || !ldh_ok(lddy, g->Cout) ||
        (reg_coef != 0.f && !w))
        return UNETRIR_EINVAL;
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_WGRAD), conv_flops(g), (hipStream_t)stream);
    return conv_wgrad_bf16_impl(g, (const __bf16*)x, ldx, (const __bf16*)dy, lddy, dw, reg_coef, w, ws, ws_bytes, (hipStream_t)stream);
}

namespace {
struct SinkScope {         // the weight-gradient launchers below this scope record their reduction in *d instead of launching it
    explicit SinkScope(unetrir_reduce_desc* d) { d->part = nullptr; d->nsplit = 0; d->n = 0; d->out = nullptr; d->reg = 0.f; d->w = nullptr; set_reduce_sink(d); }
    ~SinkScope() { set_reduce_sink(nullptr); }
};
}  // namespace

int unetrir_conv2d_wgrad_partials_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* dy, int lddy,
                                       float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes, unetrir_reduce_desc* desc,
                                       unetrir_stream_t stream) {
    if (!desc) return UNETRIR_EINVAL;
    SinkScope sc(desc);
    return unetrir_conv2d_wgrad_bf16(g, x, ldx, dy, lddy, dw, reg_coef, w, ws, ws_bytes, stream);
}

int unetrir_conv2d_transpose_wgrad_partials_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* dy,
                                                 int lddy, float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes,
                                                 unetrir_reduce_desc* desc, unetrir_stream_t stream) {
    if (!desc) return UNETRIR_EINVAL;
    SinkScope sc(desc);
    return unetrir_conv2d_transpose_wgrad_bf16(g, x, ldx, dy, lddy, dw, reg_coef, w, ws, ws_bytes, stream);
}

int unetrir_conv2d_transpose_fwd_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* wt,
                                      const float* bias, unetrir_bf16* y, int ldy, unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !wt || !y || (g->Cin & 7) || !ldh_ok(ldx, g->Cin) || ldy < g->Cout)
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_FWD), conv_flops(&c), (hipStream_t)stream);
    return conv_dgrad_impl<BF16>(&c, (const __bf16*)x, ldx, (const __bf16*)wt, bias, nullptr, 0, (__bf16*)y, ldy, (hipStream_t)stream);
}

int unetrir_conv2d_transpose_dgrad_packed_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy, const unetrir_bf16* w,
                                               const unetrir_bf16* w_packed, const unetrir_bf16* addend, int ldadd,
                                               unetrir_bf16* dx, int lddx, unetrir_stream_t stream) {
    if (!geom_ok(g) || !dy || !w || !dx || (g->Cout & 7) || !ldh_ok(lddy, g->Cout) || lddx < g->Cin)
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_DGRAD), conv_flops(&c), (hipStream_t)stream);
    return conv_fwd_impl<BF16>(&c, (const __bf16*)dy, lddy, (const __bf16*)w, nullptr, (const __bf16*)addend, ldadd, (__bf16*)dx, lddx,
                               (hipStream_t)stream, nullptr, w_packed);
}

int unetrir_conv2d_transpose_dgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* dy, int lddy, const unetrir_bf16* w,
                                        const unetrir_bf16* addend, int ldadd, unetrir_bf16* dx, int lddx,
                                        unetrir_stream_t stream) {
    if (!geom_ok(g) || !dy || !w || !dx || (g->Cout & 7) || !ldh_ok(lddy, g->Cout) || lddx < g->Cin)
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_DGRAD), conv_flops(&c), (hipStream_t)stream);
    return conv_fwd_impl<BF16>(&c, (const __bf16*)dy, lddy, (const __bf16*)w, nullptr, (const __bf16*)addend, ldadd, (__bf16*)dx, lddx,
                               (hipStream_t)stream);
}

int unetrir_conv2d_transpose_wgrad_bf16(const unetrir_conv_geom* g, const unetrir_bf16* x, int ldx, const unetrir_bf16* dy,
                                        int lddy, float* dw, float reg_coef, const float* w, void* ws, size_t ws_bytes,
                                        unetrir_stream_t stream) {
    if (!geom_ok(g) || !x || !dy || !dw || (g->Cin & 7) || (g->Cout & 7) || !ldh_ok(lddy, g->Cout) ||
        !ldh_ok(ldx, g->Cin) || (reg_coef != 0.f && !w))
        return UNETRIR_EINVAL;
    const unetrir_conv_geom c = adjoint_geom(g);
    ProfScope ps(conv_family(g, UNETRIR_FAM_CONV_WGRAD), conv_flops(&c), (hipStream_t)stream);
    return conv_wgrad_bf16_impl(&c, (const __bf16*)dy, lddy, (const __bf16*)x, ldx, dw, reg_coef, w, ws, ws_bytes, (hipStream_t)stream);
}

/* fp32 master weights [N][T][C] -> bf16 work copies: same orientation with the channel dimension padded to Cp, and
 * transposed [C][T][Np] with the row dimension padded to Np (pad entries are written as zero by the first; the second
 * leaves columns >= N untouched: pre-zero the buffer once). */
int unetrir_cast_weight_bf16(const float* w, unetrir_bf16* o, int N, int T, int C, int Cp, unetrir_stream_t stream) {
    if (!w || !o || N <= 0 || T <= 0 || C <= 0 || Cp < C) return UNETRIR_EINVAL;
    return launch_cast_weight(w, o, N, T, C, Cp, (hipStream_t)stream);
}

int unetrir_transpose_cast_weight_bf16(const float* w, unetrir_bf16* wt, int N, int T, int C, int Np, unetrir_stream_t stream) {
    if (!w || !wt || N <= 0 || T <= 0 || C <= 0 || Np < N) return UNETRIR_EINVAL;
    return launch_transpose_cast_weight(w, wt, N, T, C, Np, (hipStream_t)stream);
}

int unetrir_cast_weights_batched_bf16(const unetrir_cast_desc* desc, int n_layers, unetrir_stream_t stream) {
    if (!desc || n_layers <= 0 || n_layers > 65535) return UNETRIR_EINVAL;
    return launch_cast_weights_batched(desc, n_layers, (hipStream_t)stream);
}

/* Dense(N) (dl_models/u_net.py:259) on a small batch: y[B][N] = x[B][K] . w[N][K]^T + bias, split-K so that the 134 MB
 * weight matrix streams from every CU.  The data gradient is the same call with the transposed weight copy and no bias. */
size_t unetrir_dense_fwd_ws_bytes(int B, int K, int N) { return dense_fwd_ws_bytes(B, K, N); }

int unetrir_dense_fwd_f32(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int K, int N,
                          void* ws, size_t ws_bytes, unetrir_stream_t stream) {
    if (!x || !w || !y || !ws || B <= 0 || K <= 0 || N <= 0 || (K & 3) || ldx < K || (ldx & 3) || ldy < N) return UNETRIR_EINVAL;
    return launch_dense_fwd(x, ldx, w, bias, y, ldy, B, K, N, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_dense_dgrad_supported(int B, int K, int N) { return dense_dgrad_applies(B, K, N) ? 1 : 0; }

size_t unetrir_dense_dgrad_ws_bytes(int B, int K, int N) { return dense_dgrad_applies(B, K, N) ? dense_dgrad_ws_bytes(B, K, N) : 0; }

int unetrir_dense_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx, int B, int K, int N, void* ws,
                            size_t ws_bytes, unetrir_stream_t stream) {
    if (!dy || !w || !dx || !ws || !dense_dgrad_applies(B, K, N) || lddy < N || lddx < K || ((uintptr_t)w & 15) || ((uintptr_t)ws & 15))
        return UNETRIR_EINVAL;
    return launch_dense_dgrad(dy, lddy, w, dx, lddx, B, K, N, ws, ws_bytes, (hipStream_t)stream);
}

int unetrir_transpose_weight_f32(const float* w, float* wt, int N, int T, int C, unetrir_stream_t stream) {
    if (!w || !wt || N <= 0 || T <= 0 || C <= 0) return UNETRIR_EINVAL;
    return launch_transpose_weight(w, wt, N, T, C, (hipStream_t)stream);
}

int unetrir_stage_h2d(int n, const void* const* src, void* const* pinned, void* const* dev, const size_t* bytes,
                      unetrir_stream_t stream) {
    if (n <= 0 || !src || !pinned || !dev || !bytes) return UNETRIR_EINVAL;
    for (int k = 0; k < n; ++k) {
        if (!src[k] || !pinned[k] || !dev[k]) return UNETRIR_EINVAL;
        std::memcpy(pinned[k], src[k], bytes[k]);
        const hipError_t e = hipMemcpyAsync(dev[k], pinned[k], bytes[k], hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int unetrir_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    g_prof_mask = on == 2 ? (1u << UNETRIR_FAM_CONV_FWD) : ~0u;
    if (!g_prof_on) {
        for (auto& r : g_prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
        g_prof.clear();
    }
    return 0;
}

int unetrir_prof_collect(int* counts, double* ms, double* flops) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < UNETRIR_PROF_FAMILIES; ++i) { counts[i] = 0; ms[i] = 0.0; flops[i] = 0.0; }
    for (auto& r : g_prof) {
        hipEventSynchronize(r.e1);
        float t = 0.f;
        hipEventElapsedTime(&t, r.e0, r.e1);
        if (r.fam >= 0 && r.fam < UNETRIR_PROF_FAMILIES) { counts[r.fam]++; ms[r.fam] += t; flops[r.fam] += r.flops; }
        if (r.tag >= 0 && r.tag < UNETRIR_PROF_FAMILIES) { counts[r.tag]++; ms[r.tag] += t; flops[r.tag] += r.flops; }
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return 0;
}

}  // extern "C"
