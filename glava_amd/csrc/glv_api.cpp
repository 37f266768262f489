// glv_api.cpp -- the C ABI of include/glv_spectrum.h on top of the gfx950 kernels.
//
// Host-side responsibilities only: argument validation, constant tables (window, twiddles,
// frame weights -- glv_tables.h), device state (gravity buffers, history rings, PCM rings),
// launch geometry, HIP-event timing.  All arithmetic on samples happens in the kernels;
// there is no CPU compute path here and none is ever substituted.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/glv_spectrum.h"
#include "glv_frame.h"
#include "glv_launch.h"
#include "glv_tables.h"

namespace {

thread_local std::string g_err = "";

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(GLV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int log2_exact(uint32_t v) {
    int l = 0;
    while ((1u << l) < v) ++l;
    return (1u << l) == v ? l : -1;
}

int validate(const glv_params* p) {
    if (!p) return fail(GLV_ERR_INVALID, "params is NULL");
    const int l = log2_exact(p->n);
    if (l < 8 || l > 15) return fail(GLV_ERR_INVALID, "n=%u: must be a power of two in [256, 32768]", p->n);
    if (p->channels != 1 && p->channels != 2) return fail(GLV_ERR_INVALID, "channels=%u: must be 1 or 2", p->channels);
    if (p->avg_frames < 1 || p->avg_frames > GLV_MAX_AVG_FRAMES)
        return fail(GLV_ERR_INVALID, "avg_frames=%u: must be in [1, %d]", p->avg_frames, GLV_MAX_AVG_FRAMES);
    if (p->avg_window_kind > 1) return fail(GLV_ERR_INVALID, "avg_window_kind=%u: must be 0 or 1", p->avg_window_kind);
    if (p->log_mode > 2) return fail(GLV_ERR_INVALID, "log_mode=%u: must be 0, 1 or 2", p->log_mode);
    if (p->gl_storage > 2) return fail(GLV_ERR_INVALID, "gl_storage=%u: must be 0, 1 (GL_R16 state, one launch) or 2 (pass by pass, f32 state)", p->gl_storage);
    if (p->ur != p->ur) return fail(GLV_ERR_INVALID, "ur is NaN");   // 0 is legal: render.c:2387 yields it after an interval without updates
    if (p->round_formula > GLV_ROUND_LINEAR) return fail(GLV_ERR_INVALID, "round_formula=%u: 0 sinusoidal, 1 circular, 2 linear", p->round_formula);
    if (p->sample_mode > GLV_SAMPLE_HYBRID) return fail(GLV_ERR_INVALID, "sample_mode=%u: 0 average, 1 maximum, 2 hybrid", p->sample_mode);
    return GLV_OK;
}

// smooth_audio()'s shape as the tables take it: 0 in a glv_params field is the shipped value (smooth_parameters.glsl:17-42)
float shape_scale(const glv_params& p) { return p.sample_scale != 0.0f ? p.sample_scale : 8.0f; }
float shape_range(const glv_params& p) { return p.sample_range != 0.0f ? p.sample_range : 0.9f; }
float shape_hybrid(const glv_params& p) { return p.sample_hybrid_weight != 0.0f ? p.sample_hybrid_weight : 0.65f; }
bool same_bits(float a, float b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }
bool same_shape(const glv_params& a, const glv_params& b) {
    return a.round_formula == b.round_formula && a.sample_mode == b.sample_mode && same_bits(a.sample_hybrid_weight, b.sample_hybrid_weight)
           && same_bits(a.sample_scale, b.sample_scale) && same_bits(a.sample_range, b.sample_range);
}
glv::BarShape bar_shape(const glv_params& p) { return glv::BarShape{p.round_formula, shape_scale(p), shape_range(p), p.sample_mode == GLV_SAMPLE_AVERAGE}; }

int ensure_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(GLV_ERR_NO_DEVICE, "no usable HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(GLV_ERR_INVALID, "device %d out of range [0, %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    return GLV_OK;
}

// ---- launch wisdom ----------------------------------------------------------------------------------------------------
// The role glfft's FFTWisdom plays in the reference tree (glfft/glfft_wisdom.cpp:235-446: time candidate work-group shapes
// and radix splits per transform on the target, remember the winner per transform description).  Two things are open at run
// time here: WHICH kernel configuration of the size runs (glv_inst.hip Tuned<K, V>: points per lane / radix split, rows per
// workgroup, where the window and twiddle tables live) and HOW MANY persistent workgroups a launch uses.  Both depend on the
// size, the operator chain, the stream count and the part.  glv_batch_autotune measures the candidates on the device the
// batch lives on; entries are process-wide, keyed on the device's identity (name, CU count) so that a file tuned on one part
// is not applied on another, can be saved and loaded (GLV_WISDOM=<file> loads one when the first batch is created) and are
// consulted by every launch (through a per-batch cache: no lock on the launch path once an answer is cached).
struct WisdomKey { uint32_t n, in_kind, ops_class, log_mode, streams_log2, avg_frames, cus; char device[48]; };
struct WisdomEntry { WisdomKey k; int variant, grid; float ms; };
std::mutex g_wisdom_mu;
std::vector<WisdomEntry> g_wisdom;
std::atomic<uint64_t> g_wisdom_gen{1};      // bumped by every change of the table: invalidates the per-batch caches
bool g_wisdom_env_loaded = false;

constexpr unsigned kOpGl16 = 1u << 31;     // internal: the chain runs as the fused GL_R16 kernel (gl_storage == 1)
uint32_t ops_class_of(unsigned ops) {      // the kernel instantiation a chain selects (glv_kernel_tmpl.h launch_variant)
    if (ops & kOpGl16) return (ops & GLV_OP_BARS) ? 6 : 5;
    if (ops & GLV_OP_BARS) return 2;
    if (ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) return (ops & GLV_OP_R16) ? 4 : 1;
    return (ops & GLV_OP_R16) ? 3 : 0;
}
constexpr int kOpsClasses = 7, kInKinds = 5;
uint32_t log2_round(uint32_t v) { uint32_t l = 0; while ((2u << l) <= v) ++l; return ((v >> l << l) * 3 / 2 <= v && l < 31) ? l + 1 : l; }
bool same_key(const WisdomKey& a, const WisdomKey& b) {
    return a.n == b.n && a.in_kind == b.in_kind && a.ops_class == b.ops_class && a.log_mode == b.log_mode && a.streams_log2 == b.streams_log2
           && a.avg_frames == b.avg_frames && a.cus == b.cus && std::strncmp(a.device, b.device, sizeof(a.device)) == 0;
}
void key_set_device(WisdomKey& k, const char* name, uint32_t cus) {
    std::memset(k.device, 0, sizeof(k.device));
    size_t j = 0;
    for (const char* c = name; *c && j + 1 < sizeof(k.device); ++c) k.device[j++] = (*c == ' ' || *c == '\t' || *c == '\n') ? '_' : *c;
    if (j == 0) std::strcpy(k.device, "unknown");
    k.cus = cus;
}
bool wisdom_lookup(const WisdomKey& k, int* variant, int* grid) {
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    for (const WisdomEntry& e : g_wisdom) if (same_key(e.k, k)) { *variant = e.variant; *grid = e.grid; return true; }
    return false;
}
void wisdom_store(const WisdomKey& k, int variant, int grid, float ms) {
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    g_wisdom_gen.fetch_add(1, std::memory_order_release);
    for (WisdomEntry& e : g_wisdom) if (same_key(e.k, k)) { e.variant = variant; e.grid = grid; e.ms = ms; return; }
    g_wisdom.push_back(WisdomEntry{k, variant, grid, ms});
}
// Returns the number of entries loaded, -1 when the file cannot be opened.  Lines that are not entries of the current (v2,
// 11-field) format are counted: *skipped, of which *v1 look like the 7-field format of round 2 (n input_kind ops_class log_mode
// log2(streams) workgroups ms) -- a tuned deployment must not fall back to defaults unnoticed.
int wisdom_load_file(const char* path, int* skipped = nullptr, int* v1 = nullptr) {
    FILE* f = std::fopen(path, "r");
    if (!f) { (void) fail(GLV_ERR_INVALID, "cannot open wisdom file %s", path); return -1; }
    char line[320];
    int n_loaded = 0, n_skipped = 0, n_v1 = 0;
    while (std::fgets(line, sizeof(line), f)) {
        if (line[0] == '#' || line[0] == '\n') continue;
        WisdomKey k; int variant, grid; float ms; char dev[48];
        std::memset(&k, 0, sizeof(k));
        if (std::sscanf(line, "%47s %u %u %u %u %u %u %u %d %d %f", dev, &k.cus, &k.n, &k.in_kind, &k.ops_class, &k.log_mode, &k.streams_log2,
                        &k.avg_frames, &variant, &grid, &ms) == 11 && grid > 0 && variant >= 0) {
            key_set_device(k, dev, k.cus);
            wisdom_store(k, variant, grid, ms); ++n_loaded;
        } else {
            ++n_skipped;
            unsigned a[5]; int g1; float m1; char tail[8];
            if (std::sscanf(line, "%u %u %u %u %u %d %f %7s", &a[0], &a[1], &a[2], &a[3], &a[4], &g1, &m1, tail) == 7) ++n_v1;
        }
    }
    std::fclose(f);
    if (skipped) *skipped = n_skipped;
    if (v1) *v1 = n_v1;
    return n_loaded;
}

// Device-resident constants of one transform size.  The size-only tables -- twiddles, the window in both forms (the float-pair
// form is SEARCHED on the device: n workgroups trying 65 536 sample values per position), the log table -- are made once per
// (device, n) and shared by every batch of that size with a reference count (glv_multi shards, the five configs[4]
// batches and the audio backends used to repeat the search and its synchronisation points); the tilt factors depend on a batch's
// parameters and stay per batch.
struct SharedTables {
    int device = 0; uint32_t n = 0; int refs = 0;
    glv::cf* d_tw = nullptr; double* d_win = nullptr; float* d_win_split = nullptr; glv::LogEntry* d_log = nullptr;
    int win_shifted = 0;
};
std::mutex g_tab_mu;
std::vector<SharedTables*> g_tabs;

struct Tables {
    SharedTables* shared = nullptr;
    glv::cf* d_tw = nullptr;
    double* d_win = nullptr;
    float* d_win_split = nullptr;      // the same window as float pairs, for s16 samples (glv_core.h WinSplit; made on the device)
    int win_shifted = 0;               // positions whose low part was moved by an ulp or more (diagnostics)
    glv::LogEntry* d_log = nullptr;
    float* d_tilt = nullptr;
    float tilt_scale = 0.f, tilt_cutoff = 0.f;
    bool tilt_fold = false;
    uint32_t n_ = 0;
    // (re)generate the tilt factors when fft_scale / fft_cutoff / log mode change (render.c:845)
    int set_tilt(float fft_scale, float fft_cutoff, bool fold) {
        if (d_tilt && fft_scale == tilt_scale && fft_cutoff == tilt_cutoff && fold == tilt_fold) return GLV_OK;
        std::vector<float> t(n_);
        glv::make_tilt(t.data(), n_, fft_scale, fft_cutoff, fold);
        tilt_fold = fold;
        if (!d_tilt) HIP_TRY(hipMalloc(&d_tilt, sizeof(float) * n_));
        HIP_TRY(hipMemcpy(d_tilt, t.data(), sizeof(float) * n_, hipMemcpyHostToDevice));
        tilt_scale = fft_scale; tilt_cutoff = fft_cutoff;
        return GLV_OK;
    }
    static void free_shared(SharedTables* t) {
        if (t->d_tw) (void) hipFree(t->d_tw);
        if (t->d_win) (void) hipFree(t->d_win);
        if (t->d_win_split) (void) hipFree(t->d_win_split);
        if (t->d_log) (void) hipFree(t->d_log);
        delete t;
    }
    static int make_shared(SharedTables* t) {
        const uint32_t n = t->n, nn = n / 2;
        std::vector<glv::cf> tw(nn, glv::cf{0.0f, 0.0f});
        std::vector<double> win(n);
        glv::make_twiddles(tw.data(), nn);
        glv::make_window(win.data(), n);
        HIP_TRY(hipMalloc(&t->d_tw, sizeof(glv::cf) * nn));
        HIP_TRY(hipMalloc(&t->d_win, sizeof(double) * n));
        HIP_TRY(hipMemcpy(t->d_tw, tw.data(), sizeof(glv::cf) * nn, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->d_win, win.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        {   // the split window: searched and proven on the device for every s16 sample value (glv_misc.hip)
            int* d_fs = nullptr;
            HIP_TRY(hipMalloc(&t->d_win_split, sizeof(float) * 2 * n));
            HIP_TRY(hipMalloc(&d_fs, 2 * sizeof(int)));
            HIP_TRY(hipMemset(d_fs, 0, 2 * sizeof(int)));
            hipError_t e = glv::launch_window_split(t->d_win, t->d_win_split, n, d_fs, nullptr);
            int fs[2] = {0, 0};
            if (e == hipSuccess) e = hipMemcpy(fs, d_fs, sizeof(fs), hipMemcpyDeviceToHost);
            (void) hipFree(d_fs);
            HIP_TRY(e);
            if (fs[0]) return fail(GLV_ERR_HIP, "window table of n=%u has no exact float-pair form on this host's cos()", n);
            t->win_shifted = fs[1];
        }
        glv::LogEntry lt[glv::kLogTabMaxSize];
        glv::make_log_table(lt);
        HIP_TRY(hipMalloc(&t->d_log, sizeof(lt)));
        HIP_TRY(hipMemcpy(t->d_log, lt, sizeof(lt), hipMemcpyHostToDevice));
        return GLV_OK;
    }
    int create(uint32_t n, int device) {
        n_ = n;
        std::lock_guard<std::mutex> lock(g_tab_mu);       // held across the creation: a second batch of the size waits instead of searching too
        for (SharedTables* t : g_tabs) if (t->device == device && t->n == n) { shared = t; break; }
        if (!shared) {
            SharedTables* t = new (std::nothrow) SharedTables();
            if (!t) return fail(GLV_ERR_NOMEM, "out of host memory");
            t->device = device; t->n = n;
            if (int rc = make_shared(t)) { free_shared(t); return rc; }
            g_tabs.push_back(t);
            shared = t;
        }
        ++shared->refs;
        d_tw = shared->d_tw; d_win = shared->d_win; d_win_split = shared->d_win_split; d_log = shared->d_log; win_shifted = shared->win_shifted;
        return GLV_OK;
    }
    void destroy() {
        if (d_tilt) (void) hipFree(d_tilt);
        d_tilt = nullptr;
        if (shared) {
            std::lock_guard<std::mutex> lock(g_tab_mu);
            if (--shared->refs == 0) {
                for (size_t i = 0; i < g_tabs.size(); ++i) if (g_tabs[i] == shared) { g_tabs.erase(g_tabs.begin() + (long) i); break; }
                free_shared(shared);
            }
            shared = nullptr;
        }
        d_tw = nullptr; d_win = nullptr; d_win_split = nullptr; d_log = nullptr;
    }
};

void fill_common(glv::FrameArgs& a, const glv_params& p, const Tables& t) {
    std::memset(&a, 0, sizeof(a));
    a.tw = t.d_tw; a.win = t.d_win; a.win_split = t.d_win_split; a.logtab = t.d_log; a.tilt = t.d_tilt;
    a.F = p.avg_frames; a.mono = p.channels == 1; a.avg_window = p.avg_window != 0;
    a.inv_n = 1.0f / (float) p.n;
    a.fft_scale = p.fft_scale;
    a.one_minus_cutoff = 1.0F - p.fft_cutoff;                  // render.c:845
    a.g = p.gravity_step * (1.0F / p.ur);                      // render.c:728
    a.F_as_float = (float) p.avg_frames;                       // render.c:761
    a.F_rcp = 1.0F / (float) p.avg_frames;
    glv::make_frame_weights(a.wts, p.avg_frames, p.avg_window != 0, (int) p.avg_window_kind);
    for (uint32_t f = 0; f < p.avg_frames; ++f) a.wts32[f] = (float) a.wts[f];
}

}  // namespace

// =====================================================================================================
struct glv_batch {
    glv_params p;
    uint32_t streams = 0;
    unsigned ops_mask = 0;
    int device = 0;
    int log_nn = 0;
    int num_cus = 256;
    Tables tab;
    float* d_grav = nullptr;     // [streams*2][n]      gravity state owned by the batch (gravity without average)
    const float* grav_cur = nullptr;   // where the latest gravity output lives: d_grav, or the caller's output buffer of the previous
                                 // call when the output doubles as the state (render.c:733-734; see GLV_OP_PRIVATE_STATE)
    float* d_hist = nullptr;     // [streams*2][F][n]   ring (average; doubles as gravity state)
    int16_t* d_ring = nullptr;   // [streams][n][2]     FIFO ring mode
    int grav_mode = 0;           // which buffer holds gravity's `applied`: 0 not used yet, 1 d_grav (gravity without average
                                 // in the same call), 2 the newest ring slot (gravity + average fused)
    uint32_t head = 0;           // history slot receiving the next frame
    uint32_t ring_pos = 0;       // next write position in the PCM ring, in frames
    int grid_override = 0;
    int variant_override = -1;   // kernel configuration forced by glv_batch_set_variant (-1 = wisdom / default)
    int attr_log_mode = -1;      // log mode whose kernels had their function attributes set (batch_prepare)
    int last_launches = 0;       // kernels the last process call launched
    int last_grid = 0;           // workgroups of the last frame-kernel launch
    int last_variant = 0;        // kernel configuration of the last frame-kernel launch
    char device_name[48] = "unknown";
    // what the wisdom said the last time it was asked, per (input kind, kernel class): valid while `gen` equals the table's
    // generation -- the launch path takes no lock and scans nothing once an answer is cached
    struct PlanCache { uint64_t gen = 0; int variant = 0, grid = 0; bool hit = false; } plan_cache[kInKinds][kOpsClasses];
    uint32_t rows = 0;           // channel rows the state arrays and the scratch rows were sized for (streams * 2; 1 for the single-stream drop-ins)
    bool single_row = false;
    bool unfused_bars = false;   // GLV_UNFUSED_BARS in the environment at creation (diagnostics: bars always as a second launch)
    bool state16 = false;        // gl_storage == 1 at creation: d_grav / d_hist hold uint16 texels (2 bytes per value)
    float grav_g = 0.f; uint32_t grav_sub = 0; bool grav_int = false, grav_known = false;   // the gravity step on texels (glv_tables.h gravity_r16_integer_step)
    float* d_scratch = nullptr;  // [streams*2][n] spectra feeding GLV_OP_BARS
    float* d_ring_f32 = nullptr; // [streams][n][2] interleaved f32 ring (glv_batch_ring_update_f32)
    uint32_t ring_pos_f32 = 0;
    int* d_smin = nullptr;       // transform_smooth window bounds (host generated)
    int* d_smax = nullptr;
    uint32_t smooth_asz = 0;
    uint32_t smooth_reach = 0;   // floats of a row the walk touches: max smax + 1 (>= asz)
    uint32_t smooth_window = 0;  // largest smax - smin + 1; also covers smax - t and t - smin (the ring kernel's slot reuse)
    float smooth_d = -1.f, smooth_r = -1.f;
    glv::BarDesc* d_bar_desc = nullptr;   // GLV_OP_BARS tap tables (host generated)
    float* d_bar_w = nullptr;
    glv::BarItem* d_bar_items = nullptr;    // work lists for glv_bars_kernel (32 groups per row)
    // work lists for the fused epilogue (lanes / GL groups per row), one per kernel configuration of the size (their lanes per row differ)
    static constexpr int kMaxVariants = 4;
    glv::BarItem* d_bar_fitems[kMaxVariants] = {};
    uint32_t bar_fnsteps[kMaxVariants] = {}; bool bar_fusable[kMaxVariants] = {};
    uint32_t bar_nsteps = 0;
    uint32_t bar_count = 0; float bar_factor = -1.f, bar_phase = 0.f;
    glv_params bar_shape_of{};               // the shape fields (round_formula ... sample_range) the tables were made for
    glv::BarModeBlock* d_bar_mblocks = nullptr; float* d_bar_mw = nullptr; uint32_t bar_nmblocks = 0;   // sample_mode maximum / hybrid (glv_tables.h make_bar_mode_blocks)
    glv::BarMTile* d_bar_mtiles = nullptr;  // >= 256 bars (glv_tables.h make_bar_mtiles): tiles of 32 bars, their weights in MFMA operand layout, the bars'
    float* d_bar_wt = nullptr;              // {weight sum, reciprocal}, and -- when they could be cut -- the rounds of glv_bars_rows_kernel for its LDS ring
    float* d_bar_wsum = nullptr;
    glv::BarTile* d_bar_rounds = nullptr;
    uint32_t bar_ntiles = 0, bar_nrounds = 0, bar_ring_bins = 0, bar_bins_needed = 0;      // bar_bins_needed: bins of a row the many-bars kernels sample (0: all)
    // GLV_OP_BARS_ONLY: the chain lives below bar_bins_sampled (the last bin any bar has a tap on, rounded up to 64) -- when EVERY kernel configuration
    // of the size keeps those bins alive in its live class (a compile-time share of the row, FrameGeometry::live_points; whichever configuration a
    // call runs, the bins the bars sample are maintained) and a live class exists for the chain: the GL_R16 chains have one with the bars in a second
    // launch (7) and one with the bars fused (9); the float chains only the fused one (8).
    // 0: every bin is live (no flag, or one of the conditions fails: the full chain, the same results)
    uint32_t bar_bins_sampled = 0;
    uint32_t live_bins_now = 0;                     // refreshed with the bar tables and when the batch is prepared (update_live_bins)
    uint32_t live_bins() const { return live_bins_now; }
    bool ran_live = false;                          // a live kernel class has run since creation / the last reset: the state beyond the live bins is stale
    void update_live_bins() {
        live_bins_now = 0u;
        if (!(ops_mask & GLV_OP_BARS_ONLY) || bar_bins_sampled == 0 || bar_bins_sampled >= p.n || p.gl_storage > 1u || p.log_mode == 2u) return;
        for (int v = 0; v < glv::frame_variants(log_nn); ++v)
            if ((uint32_t) glv::frame_geometry(log_nn, v).live_points * 2u < bar_bins_sampled) return;
        // (a float chain's live class is the fused one: the production configuration must take the bars; a call that runs a configuration which cannot
        // -- forced, or from the wisdom -- takes the full chain for that call: it maintains every bin, the live calls the sampled ones, the bars see no difference)
        if (p.gl_storage == 0u && (!bar_fusable[0] || unfused_bars)) return;
        live_bins_now = bar_bins_sampled;
    }
    glv::BarRowsTables rows_tables() const {
        glv::BarRowsTables t{d_bar_mtiles, bar_ntiles, d_bar_wt, d_bar_wsum, d_bar_rounds, bar_nrounds, bar_ring_bins};
        t.mode = p.sample_mode; t.hybrid_weight = p.sample_hybrid_weight != 0.0f ? p.sample_hybrid_weight : 0.65f;
        t.mblocks = d_bar_mblocks; t.nmblocks = bar_nmblocks; t.mw = d_bar_mw; t.mode_bins = bar_bins_sampled < p.n ? bar_bins_sampled : p.n;
        return t;
    }
    // the same pass over TEXEL rows (the GL chains, gl_storage != 0): exact integer arithmetic on the i8 matrix cores (glv_tables.h make_bar_itiles)
    glv::BarMTile* d_bar_itiles = nullptr; int8_t* d_bar_wq = nullptr; glv::BarIFin* d_bar_fin = nullptr; glv::BarTile* d_bar_irounds = nullptr;
    uint32_t bar_intiles = 0, bar_inrounds = 0, bar_iring_bins = 0;
    bool bar_i8_none = false;    // the integer tables could not be made for these parameters (a bar wider than any ring / P > 31): the f32 chain serves
    bool bar_i8_off = false;     // GLV_NO_BARS_I8 in the environment at creation (diagnostics: the f32 matrix-core kernel on texel rows too)
    glv::BarIRowsTables irows_tables() const { return glv::BarIRowsTables{d_bar_itiles, bar_intiles, d_bar_wq, d_bar_fin, d_bar_irounds, bar_inrounds, bar_iring_bins}; }
    bool bars_i8() const { return d_bar_irounds != nullptr && bar_inrounds != 0; }
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t ev_used = 0;
    uint64_t launches = 0;
    const char* kernel_name = "glv_frame_kernel";
};

struct glv_state {
    glv_batch* b = nullptr;      // a one-row batch (one channel of one stream)
    // Staging of the host-pointer drop-ins: one pinned, device-mapped host block.  The kernel reads the n input floats
    // straight out of it over PCIe and writes its n results (or n GL_R16 texels) straight back, so a call is one launch
    // and one stream synchronisation -- no hipMemcpy in either direction (2 x 16 KB at the default size: the copies'
    // fixed cost, not their bandwidth, was what a call spent its time on).  GLV_STAGING=copy selects the device
    // buffer + two hipMemcpyAsync of round 1 (kept for A/B in tests/test_gpu_parity.py::test_single_stream_dropin_latency).
    float* h_io = nullptr;       // host view
    float* d_io = nullptr;       // device view of h_io (mapped), or a device buffer when copy staging is selected
    uint16_t* h_tex = nullptr;
    uint16_t* d_tex = nullptr;
    float* d_seq = nullptr;      // device buffer for GLV_OP_SMOOTH: its kernel walks a row element by element, which must not happen over PCIe
    bool mapped = true;
};

namespace {

// The large state arrays (gravity store, history ring): hipMalloc -- or, for the placement experiments of profiles/r06/modes.txt (the stateful
// chains run at one of two or three speeds per PROCESS), what GLV_STATE_ALLOC asks for (read when a batch is created; diagnostics, not API):
//   vmm:<MiB>   virtual memory management: one address range, physical memory created and mapped in chunks of <MiB> MiB (0: one chunk),
//               rounded up to the device's recommended granularity
//   fine        hipExtMallocWithFlags(hipDeviceMallocFinegrained);   uncached   ... (hipDeviceMallocUncached)
struct StateAlloc {
    void* ptr = nullptr; size_t size = 0; int kind = 0;                        // 0 hipMalloc / hipExtMalloc, 1 vmm
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
static std::mutex g_state_mu;
static std::vector<StateAlloc> g_state_allocs;                                  // (vmm allocations only: what state_free must unmap)

hipError_t state_malloc(void** out, size_t bytes, int device) {
    const char* pol = std::getenv("GLV_STATE_ALLOC");
    if (!pol || !*pol || !std::strcmp(pol, "malloc")) return hipMalloc(out, bytes);
    if (!std::strcmp(pol, "fine")) return hipExtMallocWithFlags(out, bytes, hipDeviceMallocFinegrained);
    if (!std::strcmp(pol, "uncached")) return hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached);
    if (!std::strncmp(pol, "vmm:", 4)) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (e != hipSuccess) return e;
        size_t chunk = (size_t) std::atol(pol + 4) << 20;
        const size_t total = (bytes + gran - 1) / gran * gran;
        if (chunk == 0 || chunk > total) chunk = total;
        chunk = (chunk + gran - 1) / gran * gran;
        StateAlloc a; a.kind = 1; a.size = total;
        if ((e = hipMemAddressReserve(&a.ptr, total, gran, nullptr, 0)) != hipSuccess) return e;
        for (size_t off = 0; off < total; off += chunk) {
            const size_t sz = off + chunk <= total ? chunk : total - off;
            hipMemGenericAllocationHandle_t h;
            if ((e = hipMemCreate(&h, sz, &prop, 0)) != hipSuccess) return e;
            if ((e = hipMemMap(static_cast<char*>(a.ptr) + off, sz, 0, h, 0)) != hipSuccess) return e;
            a.handles.push_back(h);
        }
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        if ((e = hipMemSetAccess(a.ptr, total, &acc, 1)) != hipSuccess) return e;
        *out = a.ptr;
        std::lock_guard<std::mutex> lk(g_state_mu);
        g_state_allocs.push_back(std::move(a));
        return hipSuccess;
    }
    return hipErrorInvalidValue;
}
void state_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_state_mu);
        for (size_t i = 0; i < g_state_allocs.size(); ++i)
            if (g_state_allocs[i].ptr == p) {
                StateAlloc a = std::move(g_state_allocs[i]);
                g_state_allocs.erase(g_state_allocs.begin() + (long) i);
                (void) hipMemUnmap(a.ptr, a.size);
                for (auto h : a.handles) (void) hipMemRelease(h);
                (void) hipMemAddressFree(a.ptr, a.size);
                return;
            }
    }
    (void) hipFree(p);
}

int batch_alloc(glv_batch* b, uint32_t rows) {
    const size_t n = b->p.n;
    b->state16 = b->p.gl_storage == 1;
    const size_t esz = b->state16 ? sizeof(uint16_t) : sizeof(float);          // GL_R16 state: texels
    if ((b->ops_mask & GLV_OP_AVERAGE)) {
        const size_t bytes = esz * rows * (size_t) b->p.avg_frames * n;
        HIP_TRY(state_malloc(reinterpret_cast<void**>(&b->d_hist), bytes, b->device));
        HIP_TRY(hipMemset(b->d_hist, 0, bytes));
    }
    if ((b->ops_mask & GLV_OP_GRAVITY)) {
        // kept even when the ring also exists: the single-op glv_gravity() drop-in owns its own
        // `applied` buffer exactly like the reference's separate udata slot (render.c:724)
        const size_t bytes = esz * rows * n;
        HIP_TRY(state_malloc(reinterpret_cast<void**>(&b->d_grav), bytes, b->device));
        HIP_TRY(hipMemset(b->d_grav, 0, bytes));
        b->grav_cur = b->d_grav;
    }
    // the device rings of the FIFO / PulseAudio modes, when the creation mask announces them
    if (b->ops_mask & GLV_OP_RING_S16) {
        const size_t bytes = sizeof(int16_t) * 2 * n * b->streams;
        HIP_TRY(hipMalloc(&b->d_ring, bytes));
        HIP_TRY(hipMemset(b->d_ring, 0, bytes));          // == the calloc'd rings of glava.c:487-494
    }
    if (b->ops_mask & GLV_OP_RING_F32) {
        const size_t bytes = sizeof(float) * 2 * n * b->streams;
        HIP_TRY(hipMalloc(&b->d_ring_f32, bytes));
        HIP_TRY(hipMemset(b->d_ring_f32, 0, bytes));
    }
    return GLV_OK;
}

WisdomKey wisdom_key(const glv_batch* b, int in_mode, unsigned ops) {
    WisdomKey k;
    std::memset(&k, 0, sizeof(k));
    k.n = b->p.n; k.in_kind = (uint32_t) in_mode; k.ops_class = ops_class_of(ops); k.log_mode = b->p.log_mode;
    k.streams_log2 = log2_round(b->streams);
    k.avg_frames = (ops & GLV_OP_AVERAGE) ? b->p.avg_frames : 0;          // F changes what a stateful launch moves
    key_set_device(k, b->device_name, (uint32_t) b->num_cus);
    return k;
}

// workgroups of a launch of configuration `variant` over `units` rows when nothing has been tuned
int default_grid(const glv_batch* b, uint32_t units, int variant) {
    const glv::FrameGeometry g = glv::frame_geometry(b->log_nn, variant);
    const uint32_t wgs = (units + g.rows_per_trip - 1) / g.rows_per_trip;
    // persistent workgroups: up to g.rounds rounds of what fits the chip (N=8192, 8192 streams: 0.194 ms with 256-512
    // workgroups, 0.224 ms with 2048; N<=4096, 65536 streams: eight rounds beat two by 1-4 %, profiles/r04/grid_ab.txt)
    const uint32_t round = (uint32_t) b->num_cus * (uint32_t) g.resident;
    if (wgs <= round) return (int) wgs;
    // more rounds than one even out the tail (the hardware dispatches the next workgroup to whichever CU is free), as long as a
    // workgroup still makes at least eight trips: every workgroup pays a prologue (window / table staging, pipeline fill) --
    // N=4096, 16384 streams: 0.168 ms with 512-1024 workgroups, 0.173 with 2048, 0.183 with 4096
    uint32_t grid = wgs / 8u;
    if (grid < round) grid = round;
    if (grid > (uint32_t) g.rounds * round) grid = (uint32_t) g.rounds * round;
    return (int) grid;
}

// (kernel configuration, workgroups) of the next frame-kernel launch: explicit overrides, else the wisdom, else the defaults
void launch_plan(glv_batch* b, uint32_t units, int in_mode, unsigned ops, int* variant, int* grid) {
    int v = 0, g = 0;
    const uint32_t cls = ops_class_of(ops);
    if (b->variant_override < 0 || b->grid_override <= 0) {
        glv_batch::PlanCache& pc = b->plan_cache[in_mode][cls];
        const uint64_t gen = g_wisdom_gen.load(std::memory_order_acquire);
        if (pc.gen != gen) {
            pc.hit = wisdom_lookup(wisdom_key(b, in_mode, ops), &pc.variant, &pc.grid);
            // the ring mode runs the frame mode's kernel with a rotated read position: what was tuned for frames (the input
            // glv_batch_autotune measures) serves it until an entry of its own exists
            if (!pc.hit && in_mode == glv::IN_S16_RING) pc.hit = wisdom_lookup(wisdom_key(b, glv::IN_S16_STEREO, ops), &pc.variant, &pc.grid);
            pc.gen = gen;
        }
        if (pc.hit) { v = pc.variant; g = pc.grid; }
    }
    if (b->variant_override >= 0) { if (b->variant_override != v) g = 0; v = b->variant_override; }
    if (!glv::frame_variant_ok(b->log_nn, in_mode, (int) b->p.log_mode, v)) { v = 0; g = 0; }     // not built for this input / log mode
    if (b->grid_override > 0) g = b->grid_override;
    else if (g > 0) {
        const int rpt = glv::frame_geometry(b->log_nn, v).rows_per_trip;
        const uint32_t wgs_max = (units + rpt - 1) / rpt;
        if ((uint32_t) g > wgs_max) g = (int) wgs_max;
    } else g = default_grid(b, units, v);
    *variant = v; *grid = g;
}

int timed_launch_begin(glv_batch* b, hipStream_t st) {
    if (!b->timing) return GLV_OK;
    if (b->ev_used + 2 > b->ev.size()) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        b->ev.push_back(e0); b->ev.push_back(e1);
    }
    HIP_TRY(hipEventRecord(b->ev[b->ev_used], st));
    return GLV_OK;
}
int timed_launch_end(glv_batch* b, hipStream_t st) {
    if (!b->timing) return GLV_OK;
    HIP_TRY(hipEventRecord(b->ev[b->ev_used + 1], st));
    b->ev_used += 2;
    b->launches += 1;
    return GLV_OK;
}

int ensure_smooth_tables(glv_batch* b) {
    if (b->d_smin && b->smooth_d == b->p.smooth_distance && b->smooth_r == b->p.smooth_ratio) return GLV_OK;
    if (!(b->p.smooth_ratio >= 1.0f)) return fail(GLV_ERR_INVALID, "smooth_ratio must be >= 1");
    const size_t n = b->p.n;
    std::vector<int> lo(n), hi(n);
    const size_t asz = glv::make_smooth_bounds(lo.data(), hi.data(), n, b->p.smooth_distance, b->p.smooth_ratio);
    if (!b->d_smin) { HIP_TRY(hipMalloc(&b->d_smin, sizeof(int) * n)); HIP_TRY(hipMalloc(&b->d_smax, sizeof(int) * n)); }
    HIP_TRY(hipMemcpy(b->d_smin, lo.data(), sizeof(int) * asz, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_smax, hi.data(), sizeof(int) * asz, hipMemcpyHostToDevice));
    size_t reach = asz;
    long window = 1;
    for (size_t t = 0; t < asz; ++t) {
        if ((size_t) hi[t] + 1 > reach) reach = (size_t) hi[t] + 1;
        // the span the ring must keep around step t: from the first tap (or t itself) to the last tap (or t itself)
        const long a = lo[t] < (long) t ? lo[t] : (long) t, z = hi[t] > (long) t ? hi[t] : (long) t;
        if (z - a + 1 > window) window = z - a + 1;
    }
    b->smooth_window = (uint32_t) window;
    b->smooth_reach = (uint32_t) (reach < n ? reach : n);
    b->smooth_asz = (uint32_t) asz; b->smooth_d = b->p.smooth_distance; b->smooth_r = b->p.smooth_ratio;
    return GLV_OK;
}

// GLV_OP_BARS tables: taps, weights, the work lists of glv_bars_kernel and one fused work list per kernel configuration of the
// size (their lanes per row differ).  Host generation + synchronous upload: creation / glv_batch_set_params only.
int ensure_bar_tables(glv_batch* b) {
    const bool averaging = b->p.sample_mode == GLV_SAMPLE_AVERAGE;                                       // maximum / hybrid: glv_bars_mode_kernel, no matrix-core / fused form
    const bool want_i8 = b->p.gl_storage != 0 && b->p.bars >= glv::kBarSeqMin && !b->bar_i8_off && averaging;      // chains whose rows are texels
    if (b->d_bar_desc && b->bar_count == b->p.bars && b->bar_factor == b->p.smooth_factor && b->bar_phase == b->p.bar_phase && same_shape(b->bar_shape_of, b->p)
        && (want_i8 == (b->d_bar_itiles != nullptr) || b->bar_i8_none)) return GLV_OK;
    if (b->p.bars == 0 || b->p.bars > b->p.n) return fail(GLV_ERR_INVALID, "bars=%u out of range", b->p.bars);
    {   // the shape: scale_audio(1) = -log(1 - SAMPLE_RANGE) / SAMPLE_SCALE is the last position smooth_audio() samples (a share of the row)
        const float sc = shape_scale(b->p), rg = shape_range(b->p), hw = shape_hybrid(b->p);
        if (!(sc > 0.0f && sc <= 1e6f) || !(rg > 0.0f && rg < 1.0f) || !(-logf(1.0f - rg) / sc <= 1.0f))
            return fail(GLV_ERR_INVALID, "sample_scale=%g sample_range=%g: need scale > 0, 0 < range < 1 and -log(1 - range) / scale <= 1 (smooth_audio() would fetch texels beyond the texture)", (double) sc, (double) rg);
        if (!(hw > 0.0f && hw <= 1.0f)) return fail(GLV_ERR_INVALID, "sample_hybrid_weight=%g: must be in (0, 1]", (double) hw);
    }
    if (!(b->p.smooth_factor >= 0.0f && b->p.smooth_factor <= 1.0f))       // also rejects NaN
        return fail(GLV_ERR_INVALID, "smooth_factor=%g: must be in [0, 1] (a bar would have no taps)", (double) b->p.smooth_factor);
    if (!(b->p.bar_phase >= 0.0f && b->p.bar_phase < 1.0f)) return fail(GLV_ERR_INVALID, "bar_phase=%g: must be in [0, 1)", (double) b->p.bar_phase);
    std::vector<glv::BarDesc> desc;
    std::vector<float> w;
    glv::make_bar_taps(desc, w, b->p.n, b->p.bars, b->p.smooth_factor, b->p.bar_phase, bar_shape(b->p));
    if (!glv::bar_chunks_in_row(desc, b->p.n)) return fail(GLV_ERR_INVALID, "bars: a tap chunk would leave the row (n=%u smooth_factor=%g)", b->p.n, (double) b->p.smooth_factor);
    auto drop = [](auto*& ptr) { if (ptr) { (void) hipFree(ptr); ptr = nullptr; } };
    drop(b->d_bar_desc); drop(b->d_bar_w); drop(b->d_bar_items); drop(b->d_bar_mtiles); drop(b->d_bar_wt); drop(b->d_bar_wsum); drop(b->d_bar_rounds);
    drop(b->d_bar_itiles); drop(b->d_bar_wq); drop(b->d_bar_fin); drop(b->d_bar_irounds); drop(b->d_bar_mblocks); drop(b->d_bar_mw);
    b->bar_nmblocks = 0;
    b->bar_intiles = 0; b->bar_inrounds = 0; b->bar_iring_bins = 0; b->bar_i8_none = false;
    for (int v = 0; v < glv_batch::kMaxVariants; ++v) { drop(b->d_bar_fitems[v]); b->bar_fusable[v] = false; b->bar_fnsteps[v] = 0; }
    b->bar_count = 0;
    // work lists: 256 / GL groups per row for glv_bars_kernel; T / GL groups for the frame kernel (GL = bar_lanes_of(n); fused bars:
    // whole waves per row, fewer than 2 * lanes bars).  one chunk of zero weights appended for padding items.
    const uint32_t zero_off = (uint32_t) w.size();
    const uint32_t chunk = glv::bar_chunk_of(b->p.n), gl = (uint32_t) glv::bar_lanes_of(b->p.n);
    w.resize(w.size() + chunk, 0.0f);
    std::vector<glv::BarItem> items;
    b->bar_nsteps = glv::make_bar_items(items, desc, 256 / gl, zero_off, chunk);
    HIP_TRY(hipMalloc(&b->d_bar_items, sizeof(glv::BarItem) * items.size()));
    HIP_TRY(hipMemcpy(b->d_bar_items, items.data(), sizeof(glv::BarItem) * items.size(), hipMemcpyHostToDevice));
    const int nv = glv::frame_variants(b->log_nn);
    for (int v = 0; v < nv && v < glv_batch::kMaxVariants; ++v) {
        const glv::FrameGeometry geo = glv::frame_geometry(b->log_nn, v);
        // bar totals + the dump slot fit the 2 * lanes floats of slack behind the row in LDS
        // (from 256 bars up a bar is one fma chain, glv_tables.h make_bar_mtiles: the chunked loop of the epilogue does not apply)
        b->bar_fusable[v] = geo.lanes % 64 == 0 && geo.nbuf == 1 && b->p.bars + 1 <= 2 * (uint32_t) geo.lanes && b->p.bars < glv::kBarSeqMin && averaging;
        if (!b->bar_fusable[v]) continue;
        std::vector<glv::BarItem> fitems;
        b->bar_fnsteps[v] = glv::make_bar_items(fitems, desc, (uint32_t) geo.lanes / gl, zero_off, chunk, (uint32_t) geo.bar_batch);
        HIP_TRY(hipMalloc(&b->d_bar_fitems[v], sizeof(glv::BarItem) * fitems.size()));
        HIP_TRY(hipMemcpy(b->d_bar_fitems[v], fitems.data(), sizeof(glv::BarItem) * fitems.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&b->d_bar_desc, sizeof(glv::BarDesc) * desc.size()));
    HIP_TRY(hipMalloc(&b->d_bar_w, sizeof(float) * w.size()));
    HIP_TRY(hipMemcpy(b->d_bar_desc, desc.data(), sizeof(glv::BarDesc) * desc.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_bar_w, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
    b->bar_count = b->p.bars; b->bar_factor = b->p.smooth_factor; b->bar_phase = b->p.bar_phase; b->bar_shape_of = b->p;
    b->bar_bins_sampled = 0;
    for (const glv::BarDesc& d : desc) b->bar_bins_sampled = d.first_bin + d.count > b->bar_bins_sampled ? d.first_bin + d.count : b->bar_bins_sampled;
    b->bar_bins_sampled = (b->bar_bins_sampled + 63u) & ~63u;                      // whole store instructions (the ring fill's 16-byte loads)
    // many bars (the pre-smoothing pass): tiles of 32 bars for the chain kernels; rounds for the smallest LDS ring that takes them
    b->bar_ntiles = 0; b->bar_nrounds = 0; b->bar_ring_bins = 0; b->bar_bins_needed = 0;
    b->update_live_bins();
    if (!averaging) {                          // sample_mode maximum / hybrid: one lane per bar and row off block-transposed weights, any number of bars
        std::vector<glv::BarModeBlock> blocks;
        std::vector<float> mw;
        glv::make_bar_mode_blocks(blocks, mw, desc, w);
        if (mw.empty()) mw.push_back(0.0f);
        HIP_TRY(hipMalloc(&b->d_bar_mblocks, sizeof(glv::BarModeBlock) * blocks.size()));
        HIP_TRY(hipMemcpy(b->d_bar_mblocks, blocks.data(), sizeof(glv::BarModeBlock) * blocks.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&b->d_bar_mw, sizeof(float) * mw.size()));
        HIP_TRY(hipMemcpy(b->d_bar_mw, mw.data(), sizeof(float) * mw.size(), hipMemcpyHostToDevice));
        b->bar_nmblocks = (uint32_t) blocks.size();
        if (b->bar_bins_sampled == 0) b->bar_bins_sampled = 64u;                   // (no bar has a tap: every bar is 0, or 0 / 0 in the hybrid)
        b->bar_bins_needed = b->bar_bins_sampled;                                  // what a transform in front of the bars has to store of a row
        return GLV_OK;
    }
    if (b->p.bars >= glv::kBarSeqMin) {
        std::vector<glv::BarMTile> mtiles;
        std::vector<glv::BarTile> rounds;
        std::vector<float> wt, wsum;
        for (uint32_t bins : {160u, 288u, 448u, 832u}) {                          // glv_misc.hip launch_bars: the ring sizes the kernel is built for
            if (!glv::make_bar_mtiles(mtiles, wt, wsum, rounds, desc, w, b->p.n, bins, 4u))       // 4 = glv_misc.hip kRowsWaves
                return fail(GLV_ERR_INVALID, "bars: no tile table (bars=%u)", b->p.bars);
            if (!rounds.empty()) { b->bar_ring_bins = bins; break; }
        }
        if (std::getenv("GLV_NO_BARS_ROWS")) rounds.clear();                        // (diagnostics: the one-lane-per-bar kernel for every row count)
        HIP_TRY(hipMalloc(&b->d_bar_mtiles, sizeof(glv::BarMTile) * mtiles.size()));
        HIP_TRY(hipMemcpy(b->d_bar_mtiles, mtiles.data(), sizeof(glv::BarMTile) * mtiles.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&b->d_bar_wt, sizeof(float) * wt.size()));
        HIP_TRY(hipMemcpy(b->d_bar_wt, wt.data(), sizeof(float) * wt.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&b->d_bar_wsum, sizeof(float) * wsum.size()));
        HIP_TRY(hipMemcpy(b->d_bar_wsum, wsum.data(), sizeof(float) * wsum.size(), hipMemcpyHostToDevice));
        b->bar_ntiles = (uint32_t) mtiles.size();
        b->bar_bins_needed = 0;
        for (const glv::BarDesc& d : desc) b->bar_bins_needed = d.first_bin + d.count > b->bar_bins_needed ? d.first_bin + d.count : b->bar_bins_needed;
        b->bar_bins_needed = (b->bar_bins_needed + 63u) & ~63u;                     // whole store instructions (and slack for the fill's 16-byte loads)
        if (!rounds.empty()) {
            HIP_TRY(hipMalloc(&b->d_bar_rounds, sizeof(glv::BarTile) * rounds.size()));
            HIP_TRY(hipMemcpy(b->d_bar_rounds, rounds.data(), sizeof(glv::BarTile) * rounds.size(), hipMemcpyHostToDevice));
            b->bar_nrounds = (uint32_t) rounds.size();
            const glv::BarRowsTables rt = b->rows_tables();
            HIP_TRY(glv::prepare_bars_rows(b->p.n, &rt));
        }
        // texel rows: the integer tables, for the smallest ring that takes them (glv_misc.hip launch_bars_i8: the rings the kernel is built for)
        if (want_i8) {
            std::vector<glv::BarMTile> itiles;
            std::vector<glv::BarTile> irounds;
            std::vector<int8_t> wq;
            std::vector<glv::BarIFin> fin;
            for (uint32_t bins : {160u, 288u, 448u, 832u, 1600u}) {
                if (!glv::make_bar_itiles(itiles, wq, fin, irounds, desc, w, b->p.n, bins, 4u)) { irounds.clear(); break; }
                if (!irounds.empty()) { b->bar_iring_bins = bins; break; }
            }
            if (irounds.empty()) b->bar_i8_none = true;
            else {
                HIP_TRY(hipMalloc(&b->d_bar_itiles, sizeof(glv::BarMTile) * itiles.size()));
                HIP_TRY(hipMemcpy(b->d_bar_itiles, itiles.data(), sizeof(glv::BarMTile) * itiles.size(), hipMemcpyHostToDevice));
                HIP_TRY(hipMalloc(&b->d_bar_wq, wq.size()));
                HIP_TRY(hipMemcpy(b->d_bar_wq, wq.data(), wq.size(), hipMemcpyHostToDevice));
                HIP_TRY(hipMalloc(&b->d_bar_fin, sizeof(glv::BarIFin) * fin.size()));
                HIP_TRY(hipMemcpy(b->d_bar_fin, fin.data(), sizeof(glv::BarIFin) * fin.size(), hipMemcpyHostToDevice));
                HIP_TRY(hipMalloc(&b->d_bar_irounds, sizeof(glv::BarTile) * irounds.size()));
                HIP_TRY(hipMemcpy(b->d_bar_irounds, irounds.data(), sizeof(glv::BarTile) * irounds.size(), hipMemcpyHostToDevice));
                b->bar_intiles = (uint32_t) itiles.size(); b->bar_inrounds = (uint32_t) irounds.size();
                const glv::BarIRowsTables irt = b->irows_tables();
                HIP_TRY(glv::prepare_bars_i8(b->p.n, &irt));
            }
        }
    }
    return GLV_OK;
}

// the gravity step on texels (only the GL_R16 state needs it: 65 536 evaluations on the host whenever g changes -- for the
// single-stream drop-ins that is whenever the host's measured `ur` changes, i.e. every frame)
void update_gravity_step(glv_batch* b) {
    const float g = b->p.gravity_step * (1.0F / b->p.ur);                      // render.c:728
    if (b->state16 && (!b->grav_known || std::memcmp(&g, &b->grav_g, sizeof(g)) != 0)) {
        b->grav_int = glv::gravity_r16_integer_step(g, &b->grav_sub);
        b->grav_g = g; b->grav_known = true;
    }
}

bool same_params(const glv_params& a, const glv_params& b) {
    return a.n == b.n && a.channels == b.channels && same_bits(a.fft_scale, b.fft_scale) && same_bits(a.fft_cutoff, b.fft_cutoff)
           && same_bits(a.gravity_step, b.gravity_step) && same_bits(a.ur, b.ur) && a.avg_frames == b.avg_frames && a.avg_window == b.avg_window
           && a.avg_window_kind == b.avg_window_kind && a.log_mode == b.log_mode && a.bars == b.bars && same_bits(a.smooth_factor, b.smooth_factor)
           && same_bits(a.smooth_distance, b.smooth_distance) && same_bits(a.smooth_ratio, b.smooth_ratio) && a.gl_storage == b.gl_storage
           && same_bits(a.bar_phase, b.bar_phase) && same_shape(a, b);
}

// Everything the process calls need besides the state arrays, made from b->p: tilt table, the gravity step on texels, and -- as
// announced by the creation mask -- bar tables, smooth bounds, the internal spectra rows.  Called by creation and by
// glv_batch_set_params (and by the single-stream drop-ins when their caller changes a knob): the ONLY place that allocates or
// copies synchronously; glv_batch_process_* / ring updates never do (tests/test_stream_order.py greps for it).
int batch_prepare(glv_batch* b) {
    if (int rc = b->tab.set_tilt(b->p.fft_scale, b->p.fft_cutoff, b->p.log_mode == 1)) return rc;
    update_gravity_step(b);
    // Tables are cheap and always made (an operator the creation mask did not announce only fails to get them when its
    // parameters are unusable: a later call of that operator is then refused); buffers of spectrum size are made for announced
    // operators only.
    // (an unannounced operator's unusable parameters are not this call's error: glv_last_error keeps what it said before)
    {
        const std::string said = g_err;
        const int rc = ensure_smooth_tables(b);
        if (rc != GLV_OK) { if (b->ops_mask & GLV_OP_SMOOTH) return rc; g_err = said; }
    }
    {
        const std::string said = g_err;
        const int rc = ensure_bar_tables(b);
        if (rc != GLV_OK) { if (b->ops_mask & GLV_OP_BARS) return rc; g_err = said; }
    }
    if (b->ops_mask & GLV_OP_BARS) {
        // the internal spectra rows: needed whenever bars are not computed inside the transform's launch from a row in LDS
        // (stateless chains, rows whose bars do not fit the slack behind them, SMOOTH | BARS) and no state array holds the spectra
        // -- the test must cover EVERY chain process() would run unfused (its `fused_bars`): the float chain's bars as texels
        // (BARS | R16 with gl_storage 0) leave through glv_bars_kernel, the audit log (log_mode 2) takes the GL passes one by one, and
        // GLV_UNFUSED_BARS forces two launches; only the GL_R16 chain whose every kernel configuration takes the bars goes without
        // A FLOAT chain (gl_storage 0) whose every kernel configuration takes the bars goes without as well (16384 stereo streams of
        // N = 4096 would hold 512 MiB nothing reads) -- unless the creation mask announces GLV_OP_R16: its bars as GL_R16 texels leave
        // through glv_bars_kernel, from the scratch rows (the mask's R16 bit is that hint and nothing else; gravity-only chains read the
        // state).  gl_storage 2 takes the GL passes one by one and always parks the rows.
        bool all_fused = (b->ops_mask & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) != 0 && !(b->ops_mask & GLV_OP_SMOOTH)
                         && (b->p.gl_storage == 1 || (b->p.gl_storage == 0 && !(b->ops_mask & GLV_OP_R16)))
                         && b->p.log_mode != 2 && !b->unfused_bars;
        for (int v = 0; v < glv::frame_variants(b->log_nn) && v < glv_batch::kMaxVariants; ++v) all_fused = all_fused && b->bar_fusable[v];
        if (!all_fused && !b->d_scratch) HIP_TRY(hipMalloc(&b->d_scratch, sizeof(float) * (size_t) b->rows * b->p.n));
        b->update_live_bins();
    }
    // function attributes (the > 64 KiB dynamic-LDS opt-in) of every frame kernel this batch can launch: set here, once per device
    // and instantiation, so that a process call is a plain launch (launch_variant with grid 0 = attribute only; combinations
    // that are not built answer hipErrorInvalidValue, which is not an error here)
    if (b->attr_log_mode != (int) b->p.log_mode) {
        b->attr_log_mode = (int) b->p.log_mode;
        glv::FrameArgs a;
        std::memset(&a, 0, sizeof(a));
        const struct { unsigned ops; bool bars; uint32_t gl, live; } cls[] = {
            {GLV_OP_FFT, false, 0, 0}, {GLV_OP_FFT | GLV_OP_R16, false, 0, 0}, {GLV_OP_FFT | GLV_OP_GRAVITY, false, 0, 0}, {GLV_OP_FFT | GLV_OP_GRAVITY | GLV_OP_R16, false, 0, 0},
            {GLV_OP_FFT | GLV_OP_GRAVITY, true, 0, 0}, {GLV_OP_FFT | GLV_OP_GRAVITY, false, 1, 0}, {GLV_OP_FFT | GLV_OP_GRAVITY, true, 1, 0}, {GLV_OP_FFT | GLV_OP_GRAVITY, false, 1, 1},
            {GLV_OP_FFT | GLV_OP_GRAVITY, true, 0, 1}, {GLV_OP_FFT | GLV_OP_GRAVITY, true, 1, 1}};
        for (int in_mode = 0; in_mode < kInKinds; ++in_mode)
            for (int v = 0; v < glv::frame_variants(b->log_nn); ++v)
                for (const auto& c : cls) {
                    a.ops = c.ops; a.gl_storage = c.gl; a.bars_out = c.bars ? reinterpret_cast<float*>(16) : nullptr; a.live_points = c.live;
                    (void) glv::launch_frame(b->log_nn, in_mode, (int) b->p.log_mode, v, a, 0, nullptr);
                }
        (void) hipGetLastError();
    }
    // the pass-by-pass GL chain parks the transform's f32 spectra when the caller's buffer cannot take them (texel / bar outputs)
    if (b->p.gl_storage == 2 && (b->ops_mask & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) && !b->d_scratch && !b->single_row)
        HIP_TRY(hipMalloc(&b->d_scratch, sizeof(float) * (size_t) b->rows * b->p.n));
    return GLV_OK;
}

// Argument checks shared by every batched entry point (ring updates run them BEFORE touching the ring, so that a
// rejected call leaves the ring where the caller saw it).
int check_ops(const glv_batch* b, unsigned ops, const float* d_out) {
    // gravity's output IS its new state (render.c:733-734): a chain that ends in gravity can leave the
    // spectra in the state buffer (glv_batch_gravity_state) instead of writing them a second time
    const bool state_is_output = (ops & GLV_OP_GRAVITY) && !(ops & (GLV_OP_AVERAGE | GLV_OP_SMOOTH | GLV_OP_RAW));
    if (!d_out && !(state_is_output && !(ops & GLV_OP_BARS) && !b->state16))
        return fail(GLV_ERR_INVALID, "NULL output pointer (allowed only for f32-state chains ending in gravity, see glv_batch_gravity_state)");
    const unsigned stateful = ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE);
    if (stateful & ~b->ops_mask)
        return fail(GLV_ERR_STATE, "ops 0x%x need state the batch was not created with (ops_mask 0x%x)", ops, b->ops_mask);
    if (stateful && b->state16 != (b->p.gl_storage == 1))
        return fail(GLV_ERR_STATE, "gl_storage=%u: the state of this batch was created as %s", b->p.gl_storage, b->state16 ? "GL_R16 texels (gl_storage 1)" : "floats (gl_storage 0 / 2)");
    if ((b->ops_mask & GLV_OP_BARS_ONLY) && stateful && !(ops & GLV_OP_BARS))
        return fail(GLV_ERR_STATE, "the batch was created with GLV_OP_BARS_ONLY: its state lives only below the bins the bars sample, a stateful call must ask for GLV_OP_BARS (ops 0x%x)", ops);
    if ((ops & GLV_OP_WRANGE) && (ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "GLV_OP_WRANGE excludes GLV_OP_FFT");
    if ((ops & GLV_OP_RAW) && !(ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "GLV_OP_RAW needs GLV_OP_FFT");
    if ((ops & GLV_OP_MAGNITUDE) && (ops & (GLV_OP_FFT | GLV_OP_WRANGE))) return fail(GLV_ERR_INVALID, "GLV_OP_MAGNITUDE excludes GLV_OP_FFT and GLV_OP_WRANGE");
    if (!(ops & (GLV_OP_FFT | GLV_OP_GRAVITY | GLV_OP_AVERAGE | GLV_OP_WRANGE | GLV_OP_SMOOTH | GLV_OP_MAGNITUDE | GLV_OP_R16))) return fail(GLV_ERR_INVALID, "empty ops");
    if ((ops & GLV_OP_R16) && (ops & (GLV_OP_RAW | GLV_OP_SMOOTH))) return fail(GLV_ERR_INVALID, "GLV_OP_R16 excludes GLV_OP_RAW and GLV_OP_SMOOTH");
    if ((ops & GLV_OP_R16) && !d_out) return fail(GLV_ERR_INVALID, "GLV_OP_R16 needs an output buffer");
    if ((ops & GLV_OP_OUTPUT_IS_STATE) && (!state_is_output || !d_out || (ops & (GLV_OP_BARS | GLV_OP_R16)) || b->p.gl_storage))
        return fail(GLV_ERR_INVALID, "GLV_OP_OUTPUT_IS_STATE needs a chain that ends in gravity with f32 rows out (no AVERAGE / SMOOTH / RAW / BARS / R16, gl_storage 0)");
    if ((ops & GLV_OP_BARS) && !b->d_bar_desc)
        return fail(GLV_ERR_STATE, "GLV_OP_BARS: the batch has no bar tables (bars / smooth_factor / bar_phase were unusable when it was created; tables are built at creation and by glv_batch_set_params, process calls never allocate)");
    if ((ops & GLV_OP_BARS) && (b->bar_count != b->p.bars || b->bar_factor != b->p.smooth_factor || b->bar_phase != b->p.bar_phase || !same_shape(b->bar_shape_of, b->p)))
        return fail(GLV_ERR_STATE, "GLV_OP_BARS: bar parameters changed without glv_batch_set_params");
    if ((ops & GLV_OP_SMOOTH) && (!b->d_smin || b->smooth_d != b->p.smooth_distance || b->smooth_r != b->p.smooth_ratio))
        return fail(GLV_ERR_STATE, "GLV_OP_SMOOTH: the batch has no window bounds for these parameters (unusable smooth_ratio at creation, or changed without glv_batch_set_params)");
    return GLV_OK;
}

// does `ops` run as the fused GL_R16 kernel?  (gl_storage 1, an FFT chain with state; RAW / SMOOTH / the audit log take the passes one by one)
bool gl_fused_chain(const glv_batch* b, unsigned ops) {
    return b->p.gl_storage == 1 && (ops & GLV_OP_FFT) && (ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) && !(ops & (GLV_OP_RAW | GLV_OP_SMOOTH)) && b->p.log_mode != 2;
}
// the chain as the launch plan / wisdom sees it
unsigned plan_ops(const glv_batch* b, unsigned ops) {
    if (gl_fused_chain(b, ops)) return ops | kOpGl16;
    if (b->p.gl_storage && (ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE))) return GLV_OP_FFT | (ops & GLV_OP_RAW);     // pass by pass: the frame kernel runs the transform alone
    return ops;
}

// One update of `units` channel rows through the fused kernel (or the post kernel when no FFT is asked).
// Stream-ordered: launches and asynchronous device-to-device copies only.
int process(glv_batch* b, const void* d_in, int in_mode, float* d_out, unsigned ops, uint32_t units,
            uint32_t rot, hipStream_t st) {
    if (!d_in) return fail(GLV_ERR_INVALID, "NULL device pointer");
    if (int rc = check_ops(b, ops, d_out)) return rc;
    const bool state_is_output = (ops & GLV_OP_GRAVITY) && !(ops & (GLV_OP_AVERAGE | GLV_OP_SMOOTH | GLV_OP_RAW));
    // transform_gravity keeps ONE `applied` buffer per slot (render.c:724).  Here it lives in d_grav when gravity runs
    // without average and in the newest ring slot when both run fused; a batch that mixed the two forms would silently
    // continue from a stale state, so that is refused (reset the batch, or use one batch per operator chain).
    if (ops & GLV_OP_GRAVITY) {
        const int mode = (ops & GLV_OP_AVERAGE) ? 2 : 1;
        if (b->grav_mode != 0 && b->grav_mode != mode)
            return fail(GLV_ERR_STATE, "gravity was last applied %s average on this batch and is now requested %s it: the two forms keep "
                                       "their state in different buffers (glv_batch_reset, or one batch per chain)",
                        b->grav_mode == 2 ? "fused with" : "without", mode == 2 ? "fused with" : "without");
    }
    float* d_final = d_out;
    b->last_launches = 0;
    HIP_TRY(hipSetDevice(b->device));
    const bool stateful = (ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) != 0;
    const bool gl_fused = gl_fused_chain(b, ops);                            // gl_storage 1: the GL passes are the transform's epilogue
    const bool gl_split = b->p.gl_storage != 0 && stateful && !gl_fused;     // the GL passes one by one (gl_storage 2; RAW / SMOOTH / audit log of 1)
    // which kernel configuration of this size runs, on how many workgroups (wisdom, overrides, defaults)
    int variant = 0, grid = 0;
    if (ops & GLV_OP_FFT) launch_plan(b, units, in_mode, plan_ops(b, ops), &variant, &grid);
    // GLV_OP_BARS: d_out receives the bars.  Stateful FFT chains whose rows are owned by whole waves compute
    // them inside the frame kernel from the finished row in LDS (the spectra never reach HBM, apart from
    // the state the operators keep anyway); otherwise the spectra stay internal -- in the gravity state
    // when the chain ends in gravity, in the scratch rows else -- and glv_bars_kernel runs after.
    // (the float chain's bars as GL_R16 texels leave through glv_bars_kernel; the GL_R16 chain stores them itself)
    const bool fused_bars = (ops & GLV_OP_BARS) && (ops & GLV_OP_FFT) && stateful && !(ops & GLV_OP_SMOOTH) && !gl_split
                            && (gl_fused || !(ops & GLV_OP_R16)) && variant < glv_batch::kMaxVariants && b->bar_fusable[variant]
                            && !b->unfused_bars;                         // diagnostics: force the two-kernel path
    if (ops & GLV_OP_BARS) {
        if (fused_bars || (state_is_output && !b->p.gl_storage)) d_out = nullptr;
        else {
            if (!b->d_scratch) return fail(GLV_ERR_STATE, "this GLV_OP_BARS chain needs the internal spectra rows: announce it in glv_batch_create's ops_mask (GLV_OP_BARS together with the chain's other operators; GLV_OP_R16 too when a float chain's bars are wanted as texels)");
            d_out = b->d_scratch;
        }
    }
    if (b->tab.tilt_scale != b->p.fft_scale || b->tab.tilt_cutoff != b->p.fft_cutoff || b->tab.tilt_fold != (b->p.log_mode == 1))
        return fail(GLV_ERR_STATE, "fft_scale / fft_cutoff / log_mode changed without glv_batch_set_params");
    if (ops & GLV_OP_GRAVITY) b->grav_mode = (ops & GLV_OP_AVERAGE) ? 2 : 1;

    glv::FrameArgs a;
    fill_common(a, b->p, b->tab);
    a.in = d_in; a.out = d_out; a.grav = b->grav_cur; a.grav_w = b->d_grav; a.hist = b->d_hist;
    a.units = units; a.ops = ops & ~(unsigned) (GLV_OP_PRIVATE_STATE | GLV_OP_OUTPUT_IS_STATE); a.head = b->head; a.rot = rot; a.log_mode = b->p.log_mode;
    a.grav_sub = b->grav_sub; a.grav_int = b->grav_int ? 1u : 0u;
    if (ops & GLV_OP_BARS) { a.ops &= ~(unsigned) GLV_OP_R16; a.bars_r16 = (ops & GLV_OP_R16) ? 1u : 0u; }   // with bars the texel conversion applies to the bars, the spectra stay f32
    // GLV_OP_OUTPUT_IS_STATE: a chain that ends in gravity writes ONE copy of its result (SURVEY 8d row B, 20 N bytes per frame) --
    // transform_gravity stores the same value to its `applied` array and to the buffer (render.c:733-734), so the caller's output
    // buffer can BE the state the next update reads.  Opt-in: the caller promises to leave the buffer alone until then.
    const bool gravity_only = (ops & GLV_OP_GRAVITY) && !(ops & GLV_OP_AVERAGE);
    const bool out_is_state = (ops & GLV_OP_OUTPUT_IS_STATE) != 0;           // check_ops vetted the chain
    if (out_is_state) {
        if ((const void*) d_out == d_in) return fail(GLV_ERR_INVALID, "GLV_OP_OUTPUT_IS_STATE: the output buffer must not be the input");
        a.grav_w = d_out; a.out = nullptr;
    }
    const float* grav_next = gravity_only ? (out_is_state ? d_out : b->d_grav) : b->grav_cur;
    if (fused_bars) {
        a.bar_desc = b->d_bar_desc; a.bar_items = b->d_bar_fitems[variant]; a.bar_nsteps = b->bar_fnsteps[variant]; a.bar_w = b->d_bar_w;
        a.bars = b->p.bars; a.bars_out = d_final;
    }

    hipError_t e;
    if (gl_fused) {
        // render.c:2188-2265 (+ :2277-2303 with bars) in ONE launch on uint16 state.  Bars that do not fit the row's slack in LDS
        // (bars == n: the pre-smoothing pass) sample the finished rows' floats from the scratch rows in a second launch.
        a.gl_storage = 1;
        // the rows go to the bars of a second launch and nowhere else (the scratch rows): what those bars do not sample is not stored
        if ((ops & GLV_OP_BARS) && !fused_bars && d_out == b->d_scratch && b->bar_bins_needed != 0 && b->bar_bins_needed < b->p.n)
            a.out_limit = b->bar_bins_needed * 4u;
        // GLV_OP_BARS_ONLY: ... and what they do not sample is not computed, nor is its state kept (kernel class 7; check_ops vetted the call)
        if (b->live_bins() != 0) { a.live_points = b->live_bins() / 2u; b->ran_live = true; }
        // ... and they go there as what they are, 16-bit texels (uint16 [rows][n] in the scratch rows), when the second launch is the
        // integer matrix-core pass (many bars: the pre-smoothing pass)
        const bool bars_i8 = (ops & GLV_OP_BARS) && !fused_bars && b->p.bars >= glv::kBarSeqMin && b->bars_i8();
        if (bars_i8) a.ops |= glv::OP_R16;
        if (int rc = timed_launch_begin(b, st)) return rc;
        b->last_grid = grid; b->last_variant = variant;
        e = glv::launch_frame(b->log_nn, in_mode, (int) b->p.log_mode, variant, a, grid, st); ++b->last_launches;
        if (e != hipSuccess) return fail(GLV_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        b->kernel_name = "glv_frame_kernel";
        if (ops & GLV_OP_AVERAGE) b->head = (b->head + 1) % b->p.avg_frames;
        b->grav_cur = grav_next;
        if (bars_i8) {
            const glv::BarIRowsTables irt = b->irows_tables();
            e = glv::launch_bars_i8(d_out, false, d_final, units, b->p.n, b->p.bars, &irt, st, (ops & GLV_OP_R16) != 0); ++b->last_launches;
            if (e != hipSuccess) return fail(GLV_ERR_HIP, "bars launch failed: %s", hipGetErrorString(e));
        } else if ((ops & GLV_OP_BARS) && !fused_bars) {
            const glv::BarRowsTables rt = b->rows_tables();
            e = glv::launch_bars(d_out, d_final, units, b->p.n, b->p.bars, b->bar_nsteps, b->d_bar_items, b->d_bar_desc, b->d_bar_w, st, (ops & GLV_OP_R16) != 0, &rt); ++b->last_launches;
            if (e != hipSuccess) return fail(GLV_ERR_HIP, "bars launch failed: %s", hipGetErrorString(e));
        }
        return timed_launch_end(b, st);
    }
    // many bars of rows that hold texel values as floats c / 65535 (the GL passes one by one): the same integer pass, converting back
    auto bars_of_texel_floats = [&](const float* rows) -> int {
        const glv::BarIRowsTables irt = b->irows_tables();
        const hipError_t e2 = glv::launch_bars_i8(rows, true, d_final, units, b->p.n, b->p.bars, &irt, st, (ops & GLV_OP_R16) != 0); ++b->last_launches;
        return e2 == hipSuccess ? GLV_OK : fail(GLV_ERR_HIP, "bars launch failed: %s", hipGetErrorString(e2));
    };
    const bool texel_bars_i8 = gl_split && (ops & GLV_OP_BARS) && !(ops & GLV_OP_SMOOTH) && b->p.bars >= glv::kBarSeqMin && b->bars_i8();
    // The GL twin's pass structure (render.c:2188-2265), pass by pass -- the transform first, then gravity / average as their own
    // pass over GL_R16-quantised values (glv_frame.h apply_state; state as floats with gl_storage 2, as texels with 1).  The frame
    // kernel delivers the float spectra into the caller's buffer when that is what it will hold in the end, else into the scratch rows.
    if (gl_split && (ops & GLV_OP_FFT)) {
        const bool direct = d_final && !(ops & (GLV_OP_BARS | GLV_OP_R16));
        float* d_tmp = direct ? d_final : b->d_scratch;
        if (!d_tmp) return fail(GLV_ERR_STATE, "this gl_storage chain needs the internal spectra rows (created for gl_storage 2 batches with state, and with GLV_OP_BARS in the ops_mask)");
        glv::FrameArgs a1 = a;
        a1.ops = GLV_OP_FFT | (ops & GLV_OP_RAW); a1.out = d_tmp; a1.bars_out = nullptr;     // GLV_OP_RAW: the passes then run on the raw values
        if (int rc = timed_launch_begin(b, st)) return rc;
        b->last_grid = grid; b->last_variant = variant;
        e = glv::launch_frame(b->log_nn, in_mode, (int) b->p.log_mode, variant, a1, grid, st); ++b->last_launches;
        if (e != hipSuccess) return fail(GLV_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        glv::FrameArgs a2 = a;
        a2.in = d_tmp; a2.ops = ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE | ((ops & GLV_OP_BARS) ? 0u : (unsigned) GLV_OP_R16)); a2.gl_storage = b->p.gl_storage;
        a2.out = (ops & GLV_OP_BARS) ? d_tmp : d_final;            // bars sample the finished rows; NULL = the state is the output
        a2.bars_out = nullptr;
        e = glv::launch_post(a2, b->p.n, st); ++b->last_launches;
        if (e != hipSuccess) return fail(GLV_ERR_HIP, "GL-storage pass launch failed: %s", hipGetErrorString(e));
        b->kernel_name = "glv_frame_kernel";
        if (ops & GLV_OP_AVERAGE) b->head = (b->head + 1) % b->p.avg_frames;
        b->grav_cur = grav_next;
        if (ops & GLV_OP_SMOOTH) {                 // render.c:694-718 on the finished rows (a SMOOTH chain always has an output buffer)
            e = glv::launch_smooth(a2.out, units, b->p.n, b->d_smin, b->d_smax, b->smooth_asz, b->smooth_reach, b->smooth_window, st); ++b->last_launches;
            if (e != hipSuccess) return fail(GLV_ERR_HIP, "smooth launch failed: %s", hipGetErrorString(e));
        }
        if (texel_bars_i8) {
            if (int rc = bars_of_texel_floats(d_tmp)) return rc;
        } else if (ops & GLV_OP_BARS) {
            const glv::BarRowsTables rt = b->rows_tables();
            e = glv::launch_bars(d_tmp, d_final, units, b->p.n, b->p.bars, b->bar_nsteps, b->d_bar_items, b->d_bar_desc, b->d_bar_w, st, (ops & GLV_OP_R16) != 0, &rt); ++b->last_launches;
            if (e != hipSuccess) return fail(GLV_ERR_HIP, "bars launch failed: %s", hipGetErrorString(e));
        }
        return timed_launch_end(b, st);            // the HIP-event window covers every launch of the chain
    }
    if (gl_split) a.gl_storage = b->p.gl_storage;                    // operators on planar rows: the post kernel models it directly
    // GLV_OP_BARS_ONLY on a float chain with the bars fused (kernel class 8): magnitude, state and the row in LDS for the live blocks only
    if (b->live_bins() != 0 && fused_bars && !gl_split && (ops & GLV_OP_FFT)) { a.live_points = b->live_bins() / 2u; b->ran_live = true; }

    if (int rc = timed_launch_begin(b, st)) return rc;
    const unsigned core = ops & (GLV_OP_FFT | GLV_OP_GRAVITY | GLV_OP_AVERAGE | GLV_OP_WRANGE | GLV_OP_MAGNITUDE | GLV_OP_R16);
    if (!core) {                                   // smooth / bars only: operate on a copy of the input rows
        if (in_mode != glv::IN_F32_PLANAR) return fail(GLV_ERR_INVALID, "operators without GLV_OP_FFT take planar f32 input");
        e = (const void*) d_out == d_in ? hipSuccess
            : hipMemcpyAsync(d_out, d_in, sizeof(float) * (size_t) units * b->p.n, hipMemcpyDeviceToDevice, st);
        b->kernel_name = "glv_smooth_kernel";
    } else if (ops & GLV_OP_FFT) {
        b->last_grid = grid; b->last_variant = variant;
        e = glv::launch_frame(b->log_nn, in_mode, (int) b->p.log_mode, variant, a, grid, st); ++b->last_launches;
        b->kernel_name = "glv_frame_kernel";
    } else {
        if (in_mode != glv::IN_F32_PLANAR) return fail(GLV_ERR_INVALID, "operators without GLV_OP_FFT take planar f32 input");
        e = glv::launch_post(a, b->p.n, st); ++b->last_launches;
        b->kernel_name = "glv_post_kernel";
    }
    if (e != hipSuccess) return fail(GLV_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (ops & GLV_OP_AVERAGE) b->head = (b->head + 1) % b->p.avg_frames;
    b->grav_cur = grav_next;
    if (ops & GLV_OP_SMOOTH) {                     // render.c:694-718, in place on the finished rows
        e = glv::launch_smooth(d_out, units, b->p.n, b->d_smin, b->d_smax, b->smooth_asz, b->smooth_reach, b->smooth_window, st); ++b->last_launches;
        if (e != hipSuccess) return fail(GLV_ERR_HIP, "smooth launch failed: %s", hipGetErrorString(e));
    }
    if (texel_bars_i8 && !(ops & GLV_OP_FFT) && d_out != nullptr) {      // gravity / average on planar rows with the GL storage, then many bars: texel values
        if (int rc = bars_of_texel_floats(d_out)) return rc;
    } else if ((ops & GLV_OP_BARS) && !fused_bars) {
        const glv::BarRowsTables rt = b->rows_tables();
        e = glv::launch_bars(d_out ? d_out : b->grav_cur, d_final, units, b->p.n, b->p.bars, b->bar_nsteps, b->d_bar_items, b->d_bar_desc, b->d_bar_w, st, (ops & GLV_OP_R16) != 0, &rt); ++b->last_launches;
        if (e != hipSuccess) return fail(GLV_ERR_HIP, "bars launch failed: %s", hipGetErrorString(e));
    }
    return timed_launch_end(b, st);
}

int batch_create_rows(const glv_params* p, uint32_t streams, unsigned ops_mask, int device, bool single_row, glv_batch** out) {
    if (!out) return fail(GLV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (int rc = validate(p)) return rc;
    if (streams == 0) return fail(GLV_ERR_INVALID, "streams must be > 0");
    if (streams > (1u << 30)) return fail(GLV_ERR_INVALID, "streams=%u: at most 2^30 (row indices are 32-bit)", streams);
    if (int rc = ensure_device(device)) return rc;
    {
        std::unique_lock<std::mutex> lock(g_wisdom_mu);
        const bool first = !g_wisdom_env_loaded;
        g_wisdom_env_loaded = true;
        lock.unlock();
        if (first) if (const char* w = std::getenv("GLV_WISDOM")) {                                    // a missing file is not an error
            int skipped = 0, v1 = 0;
            const int nl = wisdom_load_file(w, &skipped, &v1);
            if (nl >= 0 && skipped > 0)
                std::fprintf(stderr, "glv: GLV_WISDOM=%s: %d entries loaded, %d line(s) skipped%s\n", w, nl, skipped, v1 ? " (7-field v1 format: re-tune)" : "");
        }
    }
    glv_batch* b = new (std::nothrow) glv_batch();
    if (!b) return fail(GLV_ERR_NOMEM, "out of host memory");
    b->p = *p; b->streams = streams; b->ops_mask = ops_mask; b->device = device;
    b->rows = single_row ? 1u : streams * 2u; b->single_row = single_row;
    b->unfused_bars = std::getenv("GLV_UNFUSED_BARS") != nullptr;
    b->bar_i8_off = std::getenv("GLV_NO_BARS_I8") != nullptr;
    b->log_nn = log2_exact(p->n) - 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        b->num_cus = prop.multiProcessorCount;
        std::snprintf(b->device_name, sizeof(b->device_name), "%s", prop.gcnArchName[0] ? prop.gcnArchName : prop.name);
    }
    int rc = b->tab.create(p->n, device);
    if (rc == GLV_OK) rc = batch_alloc(b, b->rows);
    // every table and buffer the announced operators need is made here, so that the stream-ordered calls never allocate or copy
    if (rc == GLV_OK) rc = batch_prepare(b);
    if (rc != GLV_OK) { glv_batch_destroy(b); return rc; }
    *out = b;
    return GLV_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

void glv_params_default(glv_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->n = 4096;                 // shaders/glava/rc.glsl:190
    p->channels = 2;
    p->fft_scale = 10.2F;        // smooth_parameters.glsl:46 / render.c:930
    p->fft_cutoff = 0.3F;        // smooth_parameters.glsl:51 / render.c:931
    p->gravity_step = 4.2F;      // smooth_parameters.glsl:67 / render.c:911
    p->ur = 22050.0F / 256.0F;   // rate / (samplesize/4), rc.glsl:181,203; formula render.c:1674
    p->avg_frames = 5;           // smooth_parameters.glsl:56
    p->avg_window = 1;           // smooth_parameters.glsl:61
    p->avg_window_kind = 0;
    p->log_mode = 1;             // hardware log2: <= 1.8e-7 relative on every float (bar 1e-5); 0 = bit-faithful fp64
    p->bars = 80;                // radial.glsl:9 (NBARS 160, two channels)
    p->smooth_factor = 0.025F;   // smooth_parameters.glsl:72
    p->smooth_distance = 0.01F;  // render.c:917
    p->smooth_ratio = 4.0F;      // render.c:918
    p->gl_storage = 0;
    p->bar_phase = 0.0F;
}

// shared with glv_multi.cpp: record an error string for the calling thread
int glv_set_last_error(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
int glv_abi_version(void) { return GLV_ABI_VERSION; }
const char* glv_last_error(void) { return g_err.c_str(); }
int glv_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

// ---- batched -----------------------------------------------------------------------------------------
int glv_batch_create(const glv_params* p, uint32_t streams, unsigned ops_mask, int device, glv_batch** out) {
    return batch_create_rows(p, streams, ops_mask, device, false, out);
}

int glv_batch_set_params(glv_batch* b, const glv_params* p) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (int rc = validate(p)) return rc;
    if (p->n != b->p.n || p->avg_frames != b->p.avg_frames)
        return fail(GLV_ERR_STATE, "params (n=%u, F=%u) do not match the batch (n=%u, F=%u): fixed at creation", p->n, p->avg_frames, b->p.n, b->p.avg_frames);
    if ((p->gl_storage == 1) != b->state16)
        return fail(GLV_ERR_STATE, "gl_storage=%u: the state of this batch was created as %s", p->gl_storage, b->state16 ? "GL_R16 texels (gl_storage 1)" : "floats (gl_storage 0 / 2)");
    HIP_TRY(hipSetDevice(b->device));
    // the tables are rewritten in place: every kernel already queued on any stream of the device must have read them first (a
    // blocking copy from pageable memory does not order against a caller's non-blocking stream)
    HIP_TRY(hipDeviceSynchronize());
    const glv_params old = b->p;
    b->p = *p;
    int rc = batch_prepare(b);
    // a GLV_OP_BARS_ONLY batch that has run its live class keeps no state beyond the live bins: parameters under which the bars sample further than the
    // live classes keep (or that take the live class away: log_mode 2) would read state that was never maintained -- refused until the state is reset
    if (rc == GLV_OK && b->ran_live && b->live_bins() == 0)
        rc = fail(GLV_ERR_STATE, "this GLV_OP_BARS_ONLY batch has run its live kernel class: the state beyond the live bins was not kept, and these parameters "
                                 "(smooth_factor=%g bars=%u log_mode=%u) need the full chain -- glv_batch_reset first, or a new batch", (double) p->smooth_factor, p->bars, p->log_mode);
    if (rc != GLV_OK) { const std::string said = g_err; b->p = old; (void) batch_prepare(b); g_err = said; }   // a rejected change leaves the batch as it was
    for (auto& row : b->plan_cache) for (auto& pc : row) pc.gen = 0;        // log_mode is part of the wisdom key
    return rc;
}

int glv_batch_reset(glv_batch* b) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    const size_t n = b->p.n;
    size_t rows = b->rows;
    const size_t esz = b->state16 ? sizeof(uint16_t) : sizeof(float);
    if (b->d_hist) HIP_TRY(hipMemset(b->d_hist, 0, esz * rows * b->p.avg_frames * n));
    if (b->d_grav) HIP_TRY(hipMemset(b->d_grav, 0, esz * rows * n));
    if (b->d_ring) HIP_TRY(hipMemset(b->d_ring, 0, sizeof(int16_t) * 2 * n * b->streams));
    if (b->d_ring_f32) HIP_TRY(hipMemset(b->d_ring_f32, 0, sizeof(float) * 2 * n * b->streams));
    b->head = 0; b->ring_pos = 0; b->ring_pos_f32 = 0; b->grav_mode = 0; b->grav_cur = b->d_grav; b->ran_live = false;
    return GLV_OK;
}

int glv_batch_destroy(glv_batch* b) {
    if (!b) return GLV_OK;
    (void) hipSetDevice(b->device);
    b->tab.destroy();
    state_free(b->d_grav);
    state_free(b->d_hist);
    if (b->d_ring) (void) hipFree(b->d_ring);
    if (b->d_scratch) (void) hipFree(b->d_scratch);
    if (b->d_ring_f32) (void) hipFree(b->d_ring_f32);
    if (b->d_smin) (void) hipFree(b->d_smin);
    if (b->d_smax) (void) hipFree(b->d_smax);
    if (b->d_bar_desc) (void) hipFree(b->d_bar_desc);
    if (b->d_bar_w) (void) hipFree(b->d_bar_w);
    if (b->d_bar_items) (void) hipFree(b->d_bar_items);
    for (glv::BarItem* f : b->d_bar_fitems) if (f) (void) hipFree(f);
    if (b->d_bar_mtiles) (void) hipFree(b->d_bar_mtiles);
    if (b->d_bar_wt) (void) hipFree(b->d_bar_wt);
    if (b->d_bar_wsum) (void) hipFree(b->d_bar_wsum);
    if (b->d_bar_rounds) (void) hipFree(b->d_bar_rounds);
    if (b->d_bar_itiles) (void) hipFree(b->d_bar_itiles);
    if (b->d_bar_wq) (void) hipFree(b->d_bar_wq);
    if (b->d_bar_fin) (void) hipFree(b->d_bar_fin);
    if (b->d_bar_irounds) (void) hipFree(b->d_bar_irounds);
    if (b->d_bar_mblocks) (void) hipFree(b->d_bar_mblocks);
    if (b->d_bar_mw) (void) hipFree(b->d_bar_mw);
    for (hipEvent_t e : b->ev) (void) hipEventDestroy(e);
    delete b;
    return GLV_OK;
}

int glv_batch_process_s16(glv_batch* b, const int16_t* d_pcm, float* d_out, unsigned ops, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!(ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "s16 input requires GLV_OP_FFT");
    return process(b, d_pcm, glv::IN_S16_STEREO, d_out, ops, b->streams * 2, 0, (hipStream_t) hip_stream);
}

int glv_batch_process_f32(glv_batch* b, const float* d_f32, float* d_out, unsigned ops, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    return process(b, d_f32, glv::IN_F32_PLANAR, d_out, ops, b->streams * 2, 0, (hipStream_t) hip_stream);
}

int glv_batch_process_f32_stereo(glv_batch* b, const float* d_pcm, float* d_out, unsigned ops, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!(ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "interleaved input requires GLV_OP_FFT");
    return process(b, d_pcm, glv::IN_F32_STEREO, d_out, ops, b->streams * 2, 0, (hipStream_t) hip_stream);
}

// append `new_frames` frames of `fb` bytes each per stream at ring position `pos` (frames) of rings with a pitch of n frames:
// one strided copy, or two when the append wraps (any new_frames <= n: fifo.c:38,81,91 accept any sample_sz)
static int ring_append(char* d_ring, const char* d_new, uint32_t pos, uint32_t new_frames, uint32_t n, size_t fb, uint32_t streams, hipStream_t st) {
    const size_t pitch = (size_t) n * fb, width = (size_t) new_frames * fb;
    const uint32_t first = new_frames <= n - pos ? new_frames : n - pos;       // frames that fit before the wrap
    for (int part = 0; part < 2; ++part) {
        const uint32_t cnt = part == 0 ? first : new_frames - first;
        if (cnt == 0) continue;
        char* dst = d_ring + (size_t) (part == 0 ? pos : 0) * fb;
        const size_t w = (size_t) cnt * fb;
        if (d_new) HIP_TRY(hipMemcpy2DAsync(dst, pitch, d_new + (size_t) (part == 0 ? 0 : first) * fb, width, w, streams, hipMemcpyDeviceToDevice, st));
        else       HIP_TRY(hipMemset2DAsync(dst, pitch, 0, w, streams, st));   // fifo.c:67-79
    }
    return GLV_OK;
}

// the append half of a ring update: copy / zero-fill the new frames at the write position of the ring glv_batch_create
// allocated (GLV_OP_RING_S16 / _F32) and advance it.  *old_pos receives the position before the append.
static int ring_push(glv_batch* b, bool f32, const void* d_new, uint32_t new_frames, hipStream_t st, uint32_t* old_pos) {
    const uint32_t n = b->p.n;
    if (new_frames == 0 || new_frames > n)
        return fail(GLV_ERR_INVALID, "new_frames=%u: must be in [1, n=%u] (sample_sz/4 of %s)", new_frames, n, f32 ? "pulse_input.c:155-178" : "fifo.c:38,91");
    const size_t fb = f32 ? 8 : 4;
    void** ring = f32 ? reinterpret_cast<void**>(&b->d_ring_f32) : reinterpret_cast<void**>(&b->d_ring);
    uint32_t* pos = f32 ? &b->ring_pos_f32 : &b->ring_pos;
    if (!*ring)
        return fail(GLV_ERR_STATE, "the batch was created without GLV_OP_RING_%s in its ops_mask (rings are allocated at creation, ring updates never allocate)", f32 ? "F32" : "S16");
    if (int rc = ring_append(static_cast<char*>(*ring), static_cast<const char*>(d_new), *pos, new_frames, n, fb, b->streams, st)) return rc;
    *old_pos = *pos;
    *pos = (*pos + new_frames) % n;                              // the oldest frame now sits here: the window starts there
    return GLV_OK;
}

int glv_batch_ring_update_s16(glv_batch* b, const int16_t* d_new, uint32_t new_frames, float* d_out, unsigned ops,
                              void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!(ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "ring mode requires GLV_OP_FFT (glv_batch_ring_append_s16 appends without transforming)");
    hipStream_t st = (hipStream_t) hip_stream;
    HIP_TRY(hipSetDevice(b->device));
    if (int rc = check_ops(b, ops, d_out)) return rc;            // nothing is appended when the call cannot be processed
    uint32_t old_pos = 0;
    if (int rc = ring_push(b, false, d_new, new_frames, st, &old_pos)) return rc;
    const int rc = process(b, b->d_ring, glv::IN_S16_RING, d_out, ops, b->streams * 2, b->ring_pos, st);
    if (rc != GLV_OK) b->ring_pos = old_pos;                      // a failed launch leaves the ring where the caller saw it
    return rc;
}

int glv_batch_ring_update_f32(glv_batch* b, const float* d_new, uint32_t new_frames, float* d_out, unsigned ops, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!(ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "ring mode requires GLV_OP_FFT (glv_batch_ring_append_f32 appends without transforming)");
    if (!d_new) return fail(GLV_ERR_INVALID, "d_new is NULL (the PulseAudio backend has no zero-fill path)");
    hipStream_t st = (hipStream_t) hip_stream;
    HIP_TRY(hipSetDevice(b->device));
    if (int rc = check_ops(b, ops, d_out)) return rc;
    uint32_t old_pos = 0;
    if (int rc = ring_push(b, true, d_new, new_frames, st, &old_pos)) return rc;
    const int rc = process(b, b->d_ring_f32, glv::IN_F32_RING, d_out, ops, b->streams * 2, b->ring_pos_f32, st);
    if (rc != GLV_OK) b->ring_pos_f32 = old_pos;
    return rc;
}

int glv_batch_ring_append_s16(glv_batch* b, const int16_t* d_new, uint32_t new_frames, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    HIP_TRY(hipSetDevice(b->device));
    uint32_t old_pos = 0;
    return ring_push(b, false, d_new, new_frames, (hipStream_t) hip_stream, &old_pos);
}

int glv_batch_ring_append_f32(glv_batch* b, const float* d_new, uint32_t new_frames, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!d_new) return fail(GLV_ERR_INVALID, "d_new is NULL (the PulseAudio backend has no zero-fill path)");
    HIP_TRY(hipSetDevice(b->device));
    uint32_t old_pos = 0;
    return ring_push(b, true, d_new, new_frames, (hipStream_t) hip_stream, &old_pos);
}

int glv_batch_ring_planar(glv_batch* b, int f32_ring, float* d_planar, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!d_planar) return fail(GLV_ERR_INVALID, "NULL device pointer");
    const void* ring = f32_ring ? static_cast<const void*>(b->d_ring_f32) : static_cast<const void*>(b->d_ring);
    if (!ring) return fail(GLV_ERR_STATE, "the batch has no %s ring yet (create it with GLV_OP_RING_%s or append to it first)", f32_ring ? "f32" : "s16", f32_ring ? "F32" : "S16");
    HIP_TRY(hipSetDevice(b->device));
    hipError_t e = glv::launch_ring_planar(ring, f32_ring ? 1 : 0, b->p.n, f32_ring ? b->ring_pos_f32 : b->ring_pos, b->p.channels == 1, b->streams,
                                           d_planar, (hipStream_t) hip_stream);
    if (e != hipSuccess) return fail(GLV_ERR_HIP, "ring snapshot launch failed: %s", hipGetErrorString(e));
    return GLV_OK;
}

int glv_batch_gravity_state(glv_batch* b, const float** d_state) {
    if (!b || !d_state) return fail(GLV_ERR_INVALID, "NULL argument");
    if (!b->d_grav) return fail(GLV_ERR_STATE, "the batch was created without GLV_OP_GRAVITY");
    if (b->grav_mode == 2)
        return fail(GLV_ERR_STATE, "gravity runs fused with average on this batch: its state is the newest slot of the history ring "
                                   "(float [rows][F][n], not a [streams][2][n] array); request the chain's output instead");
    if (b->state16) return fail(GLV_ERR_STATE, "gl_storage 1 keeps the gravity store as uint16 texels, not floats: request the chain's output instead");
    if (b->ops_mask & GLV_OP_BARS_ONLY) return fail(GLV_ERR_STATE, "the batch was created with GLV_OP_BARS_ONLY: its gravity state exists only where the bars sample");
    *d_state = b->grav_cur;
    return GLV_OK;
}

int glv_batch_bars(glv_batch* b, const float* d_spec, float* d_bars, void* hip_stream) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!d_spec || !d_bars) return fail(GLV_ERR_INVALID, "NULL device pointer");
    if (b->p.bars == 0 || b->p.bars > b->p.n) return fail(GLV_ERR_INVALID, "bars=%u out of range", b->p.bars);
    HIP_TRY(hipSetDevice(b->device));
    if (!b->d_bar_desc) return fail(GLV_ERR_STATE, "the batch has no bar tables (bars / smooth_factor / bar_phase were unusable when it was created)");
    const glv::BarRowsTables rt = b->rows_tables();
    hipError_t e = glv::launch_bars(d_spec, d_bars, (size_t) b->streams * 2, b->p.n, b->p.bars, b->bar_nsteps, b->d_bar_items, b->d_bar_desc, b->d_bar_w,
                                    (hipStream_t) hip_stream, false, &rt);
    if (e != hipSuccess) return fail(GLV_ERR_HIP, "bars launch failed: %s", hipGetErrorString(e));
    return GLV_OK;
}

int glv_prelude_bufscale(int device, const float* d_in, float* d_out, size_t rows, uint32_t n_out, uint32_t k, void* hip_stream) {
    if (!d_in || !d_out) return fail(GLV_ERR_INVALID, "NULL device pointer");
    if (k == 0 || n_out == 0) return fail(GLV_ERR_INVALID, "bufscale k and n_out must be > 0");
    if (int rc = ensure_device(device)) return rc;
    hipError_t e = glv::launch_bufscale(d_in, d_out, rows * n_out, k, (hipStream_t) hip_stream);
    if (e != hipSuccess) return fail(GLV_ERR_HIP, "bufscale launch failed: %s", hipGetErrorString(e));
    return GLV_OK;
}

int glv_prelude_lerp(int device, const float* d_start, const float* d_end, float* d_out, size_t count, float uratio,
                     int kcounter, void* hip_stream) {
    if (!d_start || !d_end || !d_out) return fail(GLV_ERR_INVALID, "NULL device pointer");
    if (int rc = ensure_device(device)) return rc;
    float mod = uratio * (float) kcounter;               // render.c:1804-1805
    if (mod > 1.0F) mod = 1.0F;
    hipError_t e = glv::launch_lerp(d_start, d_end, d_out, count, mod, (hipStream_t) hip_stream);
    if (e != hipSuccess) return fail(GLV_ERR_HIP, "lerp launch failed: %s", hipGetErrorString(e));
    return GLV_OK;
}

int glv_batch_timing_begin(glv_batch* b) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    b->timing = true; b->ev_used = 0; b->launches = 0;
    return GLV_OK;
}

int glv_batch_timing_end(glv_batch* b, double* kernel_ms, uint64_t* launches) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    double total = 0.0;
    for (size_t i = 0; i + 1 < b->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(b->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]));
        total += ms;
    }
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = b->launches;
    b->timing = false;
    return GLV_OK;
}

uint64_t glv_batch_algorithmic_bytes(const glv_batch* b, unsigned ops, int input_is_s16) {
    if (!b) return 0;
    // SURVEY.md 8d, per stereo frame with N real samples per channel, F = avg_frames:
    //   in: 4N (s16 x 2ch) or 8N (f32 x 2ch);  out: 8N (f32), 4N (GLV_OP_R16 texels), 8 * bars (GLV_OP_BARS; 4 * bars as texels)
    //   + gravity (no average): read the state 8N, write it 8N -- and the output besides, unless the output IS the state
    //     (GLV_OP_OUTPUT_IS_STATE, or d_out == NULL: SURVEY 8d row B's 20N) or only bars leave the chip
    //   + average: read (F-1) ring slots 8N each, write the newest slot 8N (doubles as gravity state)
    //   gl_storage 1 (GL_R16 state): every state value is a 16-bit texel -- 4N per slot instead of 8N: with F = 5 and texels out
    //     4N + 16N + 4N + 4N = 28N
    //   gl_storage 2 (pass by pass): the transform's f32 spectra are written and read back by the gravity / average pass (+16N)
    const uint64_t N = b->p.n, F = b->p.avg_frames;
    const bool stateful = (ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) != 0;
    //   GLV_OP_BARS_ONLY: state traffic is counted for the live bins L only -- the bins the bars sample, NOT the (larger) share of the row the kernel
    //   class keeps (FrameGeometry::live_points: an implementation granularity, its extra bytes are traffic above the algorithmic figure)
    uint64_t L = N;
    if (b->live_bins() != 0 && stateful && (ops & GLV_OP_BARS)) L = b->live_bins();
    const uint64_t sv = (b->state16 && stateful) ? 4 * L : 8 * L;          // one state slot of both channels
    uint64_t per = input_is_s16 ? 4 * N : 8 * N;
    const bool bars = (ops & GLV_OP_BARS) != 0;
    if (bars) per += (uint64_t) ((ops & GLV_OP_R16) ? 4 : 8) * b->p.bars;
    else if (!((ops & GLV_OP_OUTPUT_IS_STATE) && !(ops & GLV_OP_AVERAGE))) per += (ops & GLV_OP_R16) ? 4 * N : 8 * N;
    if (ops & GLV_OP_AVERAGE) per += sv * (F - 1) + sv;
    else if (ops & GLV_OP_GRAVITY) per += 2 * sv;
    if (b->p.gl_storage == 2 && (ops & GLV_OP_FFT) && stateful) per += 16 * N;
    return per * b->streams;
}

uint32_t glv_batch_live_bins(const glv_batch* b) { return b ? b->live_bins() : 0u; }

const char* glv_batch_kernel_name(const glv_batch* b) { return b ? b->kernel_name : ""; }

int glv_batch_bars_arithmetic(const glv_batch* b) {
    if (!b || b->bar_count == 0 || !b->d_bar_desc) return GLV_BARS_NONE;
    if (b->p.sample_mode != GLV_SAMPLE_AVERAGE) return GLV_BARS_F32_SEQ;
    if (b->p.bars < glv::kBarSeqMin) return GLV_BARS_F32_CHAIN;
    return b->p.gl_storage != 0 && b->bars_i8() ? GLV_BARS_I8_EXACT : GLV_BARS_F32_MATRIX;
}

int glv_batch_last_grid(const glv_batch* b) { return b ? b->last_grid : 0; }
int glv_batch_last_launches(const glv_batch* b) { return b ? b->last_launches : 0; }

int glv_wisdom_clear(void) {
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    g_wisdom.clear();
    g_wisdom_gen.fetch_add(1, std::memory_order_release);
    return GLV_OK;
}
int glv_wisdom_count(void) { std::lock_guard<std::mutex> lock(g_wisdom_mu); return (int) g_wisdom.size(); }
int glv_wisdom_load(const char* path) {
    if (!path) return fail(GLV_ERR_INVALID, "path is NULL");
    int skipped = 0, v1 = 0;
    const int n = wisdom_load_file(path, &skipped, &v1);
    if (n < 0) return GLV_ERR_INVALID;
    if (skipped > 0) {
        // a file that yields nothing is an error; a partly usable one loads, with the count left in glv_last_error()
        const int code = n == 0 ? GLV_ERR_INVALID : GLV_OK;
        (void) fail(code, "wisdom file %s: %d entr%s loaded, %d line(s) skipped%s", path, n, n == 1 ? "y" : "ies", skipped,
                    v1 ? " (some are in the 7-field v1 format of an older library: re-tune with glv_batch_autotune and save again)" : " (not in the 11-field v2 format)");
        return code;
    }
    g_err = "";
    return GLV_OK;
}
int glv_wisdom_save(const char* path) {
    if (!path) return fail(GLV_ERR_INVALID, "path is NULL");
    FILE* f = std::fopen(path, "w");
    if (!f) return fail(GLV_ERR_INVALID, "cannot write wisdom file %s", path);
    std::fprintf(f, "# glv launch wisdom v2: device compute_units n input_kind ops_class log_mode log2(streams) avg_frames variant workgroups ms_per_launch\n");
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    for (const WisdomEntry& e : g_wisdom)
        std::fprintf(f, "%s %u %u %u %u %u %u %u %d %d %.6f\n", e.k.device, e.k.cus, e.k.n, e.k.in_kind, e.k.ops_class, e.k.log_mode, e.k.streams_log2,
                     e.k.avg_frames, e.variant, e.grid, (double) e.ms);
    std::fclose(f);
    return GLV_OK;
}

// Time every kernel configuration built for this batch's size (glv_inst.hip Tuned<K, V>) on a few workgroup counts each, on
// the batch's device with the caller's buffers, and remember the fastest (variant, workgroups) for this (device, size, input,
// chain, log mode, stream count).  The probe launches are real updates of every stream: stateful chains are reset afterwards.
int glv_batch_autotune(glv_batch* b, const int16_t* d_pcm, float* d_out, unsigned ops, void* hip_stream, int* best_grid, float* best_ms) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!(ops & GLV_OP_FFT)) return fail(GLV_ERR_INVALID, "autotune needs GLV_OP_FFT (it tunes the frame kernel)");
    if (int rc = check_ops(b, ops, d_out)) return rc;
    if (!d_pcm) return fail(GLV_ERR_INVALID, "NULL device pointer");
    hipStream_t st = (hipStream_t) hip_stream;
    HIP_TRY(hipSetDevice(b->device));
    const uint32_t units = b->streams * 2;
    struct Cand { int variant, grid; };
    std::vector<Cand> cand;
    for (int v = 0; v < glv::frame_variants(b->log_nn); ++v) {
        if (!glv::frame_variant_ok(b->log_nn, glv::IN_S16_STEREO, (int) b->p.log_mode, v)) continue;
        const glv::FrameGeometry geo = glv::frame_geometry(b->log_nn, v);
        const uint32_t wgs = (units + geo.rows_per_trip - 1) / geo.rows_per_trip;
        const uint32_t round = (uint32_t) b->num_cus * (uint32_t) geo.resident;
        for (uint32_t g : { round / 2, round, round * 3 / 2, round * 2, round * 4 }) {
            const int c = (int) (g < 1 ? 1 : (g > wgs ? wgs : g));
            bool dup = false;
            for (const Cand& x : cand) dup |= x.variant == v && x.grid == c;
            if (!dup) cand.push_back(Cand{v, c});
        }
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int saved_grid = b->grid_override, saved_variant = b->variant_override;
    Cand best{0, 0}; float bms = 0.f; int rc = GLV_OK;
    for (int pass = 0; pass < 2 && rc == GLV_OK; ++pass)                 // pass 0 warms the clocks up, pass 1 is measured
        for (const Cand& c : cand) {
            b->grid_override = c.grid; b->variant_override = c.variant;
            const int iters = pass == 0 ? 3 : 8;
            if (hipEventRecord(e0, st) != hipSuccess) { rc = fail(GLV_ERR_HIP, "hipEventRecord failed"); break; }
            for (int i = 0; i < iters && rc == GLV_OK; ++i) rc = process(b, d_pcm, glv::IN_S16_STEREO, d_out, ops, units, 0, st);
            if (rc != GLV_OK) break;
            float ms = 0.f;
            if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
                rc = fail(GLV_ERR_HIP, "event timing failed"); break;
            }
            ms /= (float) iters;
            if (pass == 1 && (best.grid == 0 || ms < bms)) { best = c; bms = ms; }
        }
    b->grid_override = saved_grid; b->variant_override = saved_variant;
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (rc != GLV_OK) return rc;
    if (ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE)) { if (int r2 = glv_batch_reset(b)) return r2; }
    wisdom_store(wisdom_key(b, glv::IN_S16_STEREO, plan_ops(b, ops)), best.variant, best.grid, bms);
    if (best_grid) *best_grid = best.grid;
    if (best_ms) *best_ms = bms;
    return GLV_OK;
}

// Placement wisdom (round 6; profiles/r06/modes.txt).  A stateful chain runs at one of two or three speeds, 7 - 14 % apart, and which one is a property of
// WHERE ITS STATE ARRAYS LIE in physical memory relative to the caller's output buffer: K batches of one process, created one after the other, each keep
// their own speed against the same buffers, and one batch changes speed with the output buffer it is given -- not with the allocator (hipMalloc, virtual
// memory management in one or many chunks, fine-grained: the same mix), not with the TLB (UTCL1 misses 3e4 of 2.8e8 requests in every mode), not with
// clocks, power or temperature.  So the library does what it does for the launch geometry: it measures.  The current placement is timed with the caller's
// real buffers, then up to `candidates` - 1 fresh allocations of the state arrays (the earlier ones stay allocated meanwhile, so every candidate lies on
// other frames); the fastest is kept, the others are freed, the state is reset.
int glv_batch_tune_placement(glv_batch* b, const int16_t* d_pcm, void* d_out, unsigned ops, int candidates, void* hip_stream, float* first_ms, float* best_ms) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (!(ops & GLV_OP_FFT) || !(ops & (GLV_OP_GRAVITY | GLV_OP_AVERAGE))) return fail(GLV_ERR_INVALID, "glv_batch_tune_placement tunes a stateful chain: ops needs GLV_OP_FFT and GLV_OP_GRAVITY / GLV_OP_AVERAGE");
    if (int rc = check_ops(b, ops, static_cast<float*>(d_out))) return rc;
    if (!d_pcm || !d_out) return fail(GLV_ERR_INVALID, "NULL device pointer");
    if (candidates < 1) candidates = 1;
    if (candidates > 16) candidates = 16;
    hipStream_t st = (hipStream_t) hip_stream;
    HIP_TRY(hipSetDevice(b->device));
    const uint32_t units = b->streams * 2;
    const size_t esz = b->state16 ? sizeof(uint16_t) : sizeof(float);
    const size_t hist_bytes = b->d_hist ? esz * (size_t) b->rows * b->p.avg_frames * b->p.n : 0, grav_bytes = b->d_grav ? esz * (size_t) b->rows * b->p.n : 0;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int rc = GLV_OK;
    auto measure = [&](float* ms) -> int {
        if (int r = glv_batch_reset(b)) return r;
        for (int i = 0; i < 4; ++i) if (int r = process(b, d_pcm, glv::IN_S16_STEREO, static_cast<float*>(d_out), ops, units, 0, st)) return r;
        if (hipEventRecord(e0, st) != hipSuccess) return fail(GLV_ERR_HIP, "hipEventRecord failed");
        const int iters = 10;
        for (int i = 0; i < iters; ++i) if (int r = process(b, d_pcm, glv::IN_S16_STEREO, static_cast<float*>(d_out), ops, units, 0, st)) return r;
        if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(ms, e0, e1) != hipSuccess) return fail(GLV_ERR_HIP, "event timing failed");
        *ms /= (float) iters;
        return GLV_OK;
    };
    float cur = 0.f;
    for (int i = 0; i < 24 && rc == GLV_OK; ++i) rc = process(b, d_pcm, glv::IN_S16_STEREO, static_cast<float*>(d_out), ops, units, 0, st);      // clocks up
    if (rc == GLV_OK) rc = measure(&cur);
    const float first = cur;
    std::vector<std::pair<void*, void*>> losers;                                 // (hist, grav) of the placements that lost: freed at the end
    for (int c = 1; c < candidates && rc == GLV_OK; ++c) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * (hist_bytes + grav_bytes) + ((size_t) 1 << 30)) break;      // room for this candidate and slack
        void *nh = nullptr, *ng = nullptr;
        if (hist_bytes && state_malloc(&nh, hist_bytes, b->device) != hipSuccess) { (void) hipGetLastError(); break; }
        if (grav_bytes && state_malloc(&ng, grav_bytes, b->device) != hipSuccess) { (void) hipGetLastError(); state_free(nh); break; }
        void *oh = b->d_hist, *og = b->d_grav;
        if (nh) b->d_hist = static_cast<float*>(nh);
        if (ng) b->d_grav = static_cast<float*>(ng);
        float ms = 0.f;
        rc = measure(&ms);
        if (rc == GLV_OK && ms < cur * 0.985f) { cur = ms; losers.emplace_back(oh, og); }
        else { b->d_hist = static_cast<float*>(oh); b->d_grav = static_cast<float*>(og); losers.emplace_back(nh, ng); }
    }
    (void) hipStreamSynchronize(st);
    for (auto& l : losers) { state_free(l.first); state_free(l.second); }
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    if (rc != GLV_OK) return rc;
    if (int r = glv_batch_reset(b)) return r;
    if (first_ms) *first_ms = first;
    if (best_ms) *best_ms = cur;
    return GLV_OK;
}

// tuning hook used by tools/tune.py and bench.py --grid (0 = automatic)
int glv_batch_set_grid(glv_batch* b, int grid) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    b->grid_override = grid;
    return GLV_OK;
}
int glv_batch_set_variant(glv_batch* b, int variant) {
    if (!b) return fail(GLV_ERR_INVALID, "batch is NULL");
    if (variant >= glv::frame_variants(b->log_nn))
        return fail(GLV_ERR_INVALID, "variant %d: n=%u has %d kernel configuration(s)", variant, b->p.n, glv::frame_variants(b->log_nn));
    b->variant_override = variant < 0 ? -1 : variant;
    return GLV_OK;
}
int glv_batch_variants(const glv_batch* b) { return b ? glv::frame_variants(b->log_nn) : 0; }
int glv_batch_last_variant(const glv_batch* b) { return b ? b->last_variant : 0; }
int glv_batch_window_selftest(glv_batch* b, unsigned long long* mismatches, int* shifted) {
    if (!b || !mismatches) return fail(GLV_ERR_INVALID, "glv_batch_window_selftest: NULL argument");
    HIP_TRY(hipSetDevice(b->device));
    unsigned long long* d_m = nullptr;
    HIP_TRY(hipMalloc(&d_m, sizeof(*d_m)));
    hipError_t e = hipMemset(d_m, 0, sizeof(*d_m));
    if (e == hipSuccess) e = glv::launch_window_split_check(b->tab.d_win, b->tab.d_win_split, b->p.n, d_m, nullptr);
    if (e == hipSuccess) e = hipMemcpy(mismatches, d_m, sizeof(*d_m), hipMemcpyDeviceToHost);
    (void) hipFree(d_m);
    HIP_TRY(e);
    if (shifted) *shifted = b->tab.win_shifted;
    return GLV_OK;
}

int glv_batch_describe_variant(const glv_batch* b, int variant, char* buf, size_t len) {
    if (!b || !buf || len == 0) return fail(GLV_ERR_INVALID, "NULL argument");
    if (variant < 0 || variant >= glv::frame_variants(b->log_nn)) return fail(GLV_ERR_INVALID, "variant %d out of range", variant);
    const glv::FrameGeometry g = glv::frame_geometry(b->log_nn, variant);
    std::snprintf(buf, len, "n=%u variant %d: %d points per lane, %d lanes per row, %d row(s) per workgroup, %d workgroup(s) per CU, "
                            "%d KiB LDS, twiddles %s, window %s", b->p.n, variant, 1 << g.log_e, g.lanes, g.slots, g.resident, (g.lds_bytes + 1023) / 1024,
                  g.twreg == 1 ? "in VGPRs" : g.twreg == 0 ? "through L2" : g.twreg == 4 ? (g.nbuf == 0 ? "in LDS (split exchange)" : "in LDS") : "middle passes in LDS, last pass through L2",
                  g.winlds ? "in LDS" : "through L2");
    return GLV_OK;
}

// ---- single-stream drop-ins -------------------------------------------------------------------------
int glv_state_create(const glv_params* p, int device, glv_state** out) {
    if (!out) return fail(GLV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    glv_state* s = new (std::nothrow) glv_state();
    if (!s) return fail(GLV_ERR_NOMEM, "out of host memory");
    // (GL_R16 state -- the accel path of handle_audio, glv_gl_texture -- also announces the pre-smoothing pass: one scratch row)
    const unsigned mask = GLV_OP_GRAVITY | GLV_OP_AVERAGE | GLV_OP_SMOOTH | (p && p->gl_storage == 1 && p->bars >= 1 && p->bars <= p->n ? (unsigned) GLV_OP_BARS : 0u);
    int rc = batch_create_rows(p, 1, mask, device, true, &s->b);
    if (rc == GLV_OK) {
        const char* mode = std::getenv("GLV_STAGING");
        s->mapped = !(mode && std::strcmp(mode, "copy") == 0);
        hipError_t e;
        if (s->mapped) {
            e = hipHostMalloc(reinterpret_cast<void**>(&s->h_io), sizeof(float) * p->n, hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&s->d_io), s->h_io, 0);
        } else {
            e = hipMalloc(&s->d_io, sizeof(float) * p->n);
        }
        if (e != hipSuccess) rc = fail(GLV_ERR_HIP, "staging allocation failed: %s", hipGetErrorString(e));
    }
    if (rc != GLV_OK) { glv_state_destroy(s); return rc; }
    *out = s;
    return GLV_OK;
}

int glv_state_reset(glv_state* s) {
    if (!s || !s->b) return fail(GLV_ERR_INVALID, "state is NULL");
    glv_batch* b = s->b;
    HIP_TRY(hipSetDevice(b->device));
    const size_t n = b->p.n;
    const size_t esz = b->state16 ? sizeof(uint16_t) : sizeof(float);
    HIP_TRY(hipMemset(b->d_hist, 0, esz * b->p.avg_frames * n));
    HIP_TRY(hipMemset(b->d_grav, 0, esz * n));
    b->head = 0; b->grav_mode = 0; b->grav_cur = b->d_grav;
    return GLV_OK;
}

int glv_state_destroy(glv_state* s) {
    if (!s) return GLV_OK;
    if (s->b) { (void) hipSetDevice(s->b->device); glv_batch_destroy(s->b); }
    if (s->d_seq) (void) hipFree(s->d_seq);
    if (s->mapped) {
        if (s->h_io) (void) hipHostFree(s->h_io);
        if (s->h_tex) (void) hipHostFree(s->h_tex);
    } else {
        if (s->d_io) (void) hipFree(s->d_io);
        if (s->d_tex) (void) hipFree(s->d_tex);
    }
    delete s;
    return GLV_OK;
}

static int single(const glv_params* p, glv_state* s, float* buf, unsigned ops) {
    if (!s || !s->b) return fail(GLV_ERR_INVALID, "state is NULL");
    if (!buf) return fail(GLV_ERR_INVALID, "buf is NULL");
    if (int rc = validate(p)) return rc;
    glv_batch* b = s->b;
    if (p->n != b->p.n || p->avg_frames != b->p.avg_frames)
        return fail(GLV_ERR_STATE, "params (n=%u, F=%u) do not match the state (n=%u, F=%u)", p->n, p->avg_frames, b->p.n, b->p.avg_frames);
    // scalar knobs may change between calls, exactly like gl_data fields: what depends on them is regenerated when they do
    // (compared field by field: padding bytes of a caller's struct mean nothing.  GLava's measured `ur` changes every frame
    // (render.c:2387): that touches nothing but the gravity step -- no table, no launch plan)
    if (!same_params(b->p, *p)) {
        glv_params rest = *p;
        rest.ur = b->p.ur; rest.gravity_step = b->p.gravity_step;
        if (same_params(b->p, rest)) {
            b->p.ur = p->ur; b->p.gravity_step = p->gravity_step;
            update_gravity_step(b);
        } else if (int rc = glv_batch_set_params(b, p)) return rc;
    }
    HIP_TRY(hipSetDevice(b->device));
    const size_t bytes = sizeof(float) * p->n;
    if (s->mapped && (ops & GLV_OP_SMOOTH)) {
        if (!s->d_seq) HIP_TRY(hipMalloc(&s->d_seq, bytes));
        HIP_TRY(hipMemcpyAsync(s->d_seq, buf, bytes, hipMemcpyHostToDevice, nullptr));
        if (int rc = process(b, s->d_seq, glv::IN_F32_PLANAR, s->d_seq, ops, 1, 0, nullptr)) return rc;
        HIP_TRY(hipMemcpyAsync(buf, s->d_seq, bytes, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        return GLV_OK;
    }
    if (s->mapped) {
        std::memcpy(s->h_io, buf, bytes);
        const int rc = process(b, s->d_io, glv::IN_F32_PLANAR, s->d_io, ops, 1, 0, nullptr);
        HIP_TRY(hipStreamSynchronize(nullptr));        // also after a failed launch: nothing may still be reading h_io
        if (rc) return rc;
        std::memcpy(buf, s->h_io, bytes);
        return GLV_OK;
    }
    HIP_TRY(hipMemcpyAsync(s->d_io, buf, bytes, hipMemcpyHostToDevice, nullptr));
    if (int rc = process(b, s->d_io, glv::IN_F32_PLANAR, s->d_io, ops, 1, 0, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(buf, s->d_io, bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return GLV_OK;
}

int glv_fft(const glv_params* p, glv_state* s, float* buf) { return single(p, s, buf, GLV_OP_FFT); }
int glv_gravity(const glv_params* p, glv_state* s, float* buf) { return single(p, s, buf, GLV_OP_GRAVITY); }
int glv_average(const glv_params* p, glv_state* s, float* buf) { return single(p, s, buf, GLV_OP_AVERAGE); }
int glv_wrange(const glv_params* p, glv_state* s, float* buf) { return single(p, s, buf, GLV_OP_WRANGE); }
int glv_smooth(const glv_params* p, glv_state* s, float* buf) { return single(p, s, buf, GLV_OP_SMOOTH); }
int glv_magnitude(const glv_params* p, glv_state* s, float* buf) { return single(p, s, buf, GLV_OP_MAGNITUDE); }
int glv_fft_gravity_average(const glv_params* p, glv_state* s, float* buf) {
    return single(p, s, buf, GLV_OP_FFT | GLV_OP_GRAVITY | GLV_OP_AVERAGE);
}
int glv_texels_r16(const glv_params* p, glv_state* s, const float* buf, uint16_t* texels) {
    if (!s || !s->b) return fail(GLV_ERR_INVALID, "state is NULL");
    if (!buf || !texels) return fail(GLV_ERR_INVALID, "NULL buffer");
    if (int rc = validate(p)) return rc;
    glv_batch* b = s->b;
    if (p->n != b->p.n) return fail(GLV_ERR_STATE, "params n=%u does not match the state (n=%u)", p->n, b->p.n);
    HIP_TRY(hipSetDevice(b->device));
    if (s->mapped) {
        if (!s->h_tex) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_tex), sizeof(uint16_t) * p->n, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->d_tex), s->h_tex, 0));
        }
        std::memcpy(s->h_io, buf, sizeof(float) * p->n);
        const int rc = process(b, s->d_io, glv::IN_F32_PLANAR, reinterpret_cast<float*>(s->d_tex), GLV_OP_R16, 1, 0, nullptr);
        HIP_TRY(hipStreamSynchronize(nullptr));
        if (rc) return rc;
        std::memcpy(texels, s->h_tex, sizeof(uint16_t) * p->n);
        return GLV_OK;
    }
    if (!s->d_tex) HIP_TRY(hipMalloc(&s->d_tex, sizeof(uint16_t) * p->n));
    HIP_TRY(hipMemcpyAsync(s->d_io, buf, sizeof(float) * p->n, hipMemcpyHostToDevice, nullptr));
    if (int rc = process(b, s->d_io, glv::IN_F32_PLANAR, reinterpret_cast<float*>(s->d_tex), GLV_OP_R16, 1, 0, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(texels, s->d_tex, sizeof(uint16_t) * p->n, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return GLV_OK;
}

int glv_gl_texture(const glv_params* p, glv_state* s, const float* buf, int smooth_pass, uint16_t* texels) {
    if (!s || !s->b) return fail(GLV_ERR_INVALID, "state is NULL");
    if (!buf || !texels) return fail(GLV_ERR_INVALID, "NULL buffer");
    if (int rc = validate(p)) return rc;
    glv_batch* b = s->b;
    if (p->n != b->p.n || p->avg_frames != b->p.avg_frames)
        return fail(GLV_ERR_STATE, "params (n=%u, F=%u) do not match the state (n=%u, F=%u)", p->n, p->avg_frames, b->p.n, b->p.avg_frames);
    if (p->gl_storage != 1 || !b->state16) return fail(GLV_ERR_STATE, "glv_gl_texture needs a state created with gl_storage = 1 (the GL passes' GL_R16 storage)");
    if (smooth_pass && (p->bars != p->n || !(b->ops_mask & GLV_OP_BARS)))
        return fail(GLV_ERR_INVALID, "glv_gl_texture with the pre-smoothing pass: bars must equal n (bar_phase 0.5: the pass's texel centres) when the state is created");
    if (!same_params(b->p, *p)) {
        glv_params rest = *p;
        rest.ur = b->p.ur; rest.gravity_step = b->p.gravity_step;
        if (same_params(b->p, rest)) { b->p.ur = p->ur; b->p.gravity_step = p->gravity_step; update_gravity_step(b); }
        else if (int rc = glv_batch_set_params(b, p)) return rc;
    }
    HIP_TRY(hipSetDevice(b->device));
    // render.c:2230: no averaging pass with a single frame; the chain then ends in the gravity store
    const unsigned ops = GLV_OP_FFT | GLV_OP_GRAVITY | (p->avg_frames > 1 ? (unsigned) GLV_OP_AVERAGE : 0u) | (smooth_pass ? (unsigned) GLV_OP_BARS : 0u) | GLV_OP_R16;
    const size_t in_bytes = sizeof(float) * p->n, out_bytes = sizeof(uint16_t) * p->n;
    if (s->mapped) {
        if (!s->h_tex) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_tex), out_bytes, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->d_tex), s->h_tex, 0));
        }
        std::memcpy(s->h_io, buf, in_bytes);
        const int rc = process(b, s->d_io, glv::IN_F32_PLANAR, reinterpret_cast<float*>(s->d_tex), ops, 1, 0, nullptr);
        HIP_TRY(hipStreamSynchronize(nullptr));
        if (rc) return rc;
        std::memcpy(texels, s->h_tex, out_bytes);
        return GLV_OK;
    }
    if (!s->d_tex) HIP_TRY(hipMalloc(&s->d_tex, out_bytes));
    HIP_TRY(hipMemcpyAsync(s->d_io, buf, in_bytes, hipMemcpyHostToDevice, nullptr));
    if (int rc = process(b, s->d_io, glv::IN_F32_PLANAR, reinterpret_cast<float*>(s->d_tex), ops, 1, 0, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(texels, s->d_tex, out_bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return GLV_OK;
}

int glv_device_malloc(int device, size_t bytes, void** d_ptr) {
    if (!d_ptr) return fail(GLV_ERR_INVALID, "d_ptr is NULL");
    *d_ptr = nullptr;
    if (int rc = ensure_device(device)) return rc;
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) return fail(GLV_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    return GLV_OK;
}
int glv_device_free(int device, void* d_ptr) {
    if (!d_ptr) return GLV_OK;
    if (int rc = ensure_device(device)) return rc;
    HIP_TRY(hipFree(d_ptr));
    return GLV_OK;
}
int glv_device_upload(int device, void* d_dst, const void* h_src, size_t bytes, void* hip_stream) {
    if (!d_dst || !h_src) return fail(GLV_ERR_INVALID, "NULL pointer");
    if (int rc = ensure_device(device)) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t) hip_stream));
    return GLV_OK;
}
int glv_device_download(int device, void* h_dst, const void* d_src, size_t bytes, void* hip_stream) {
    if (!h_dst || !d_src) return fail(GLV_ERR_INVALID, "NULL pointer");
    if (int rc = ensure_device(device)) return rc;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t) hip_stream));
    return GLV_OK;
}
int glv_device_sync(int device, void* hip_stream) {
    if (int rc = ensure_device(device)) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t) hip_stream));
    return GLV_OK;
}

int glv_unpack_s16(int device, const int16_t* pcm, size_t frames, int channels, float* l, float* r) {
    if (!l || !r) return fail(GLV_ERR_INVALID, "NULL output");
    if (channels != 1 && channels != 2) return fail(GLV_ERR_INVALID, "channels=%d: must be 1 or 2", channels);
    if (frames == 0) return GLV_OK;
    if (int rc = ensure_device(device)) return rc;
    int16_t* d_pcm = nullptr; float* d_l = nullptr; float* d_r = nullptr;
    int rc = GLV_OK;
    auto cleanup = [&]() { if (d_pcm) (void) hipFree(d_pcm); if (d_l) (void) hipFree(d_l); if (d_r) (void) hipFree(d_r); };
#define TRY_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(GLV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)
    TRY_C(hipMalloc(&d_pcm, frames * 4));
    TRY_C(hipMalloc(&d_l, frames * 4));
    TRY_C(hipMalloc(&d_r, frames * 4));
    if (pcm) TRY_C(hipMemcpy(d_pcm, pcm, frames * 4, hipMemcpyHostToDevice));
    else     TRY_C(hipMemset(d_pcm, 0, frames * 4));            // fifo.c:67-79
    TRY_C(glv::launch_unpack(d_pcm, frames, channels == 1, d_l, d_r, nullptr));
    TRY_C(hipMemcpy(l, d_l, frames * 4, hipMemcpyDeviceToHost));
    TRY_C(hipMemcpy(r, d_r, frames * 4, hipMemcpyDeviceToHost));
#undef TRY_C
    cleanup();
    return GLV_OK;
}

}  // extern "C"
