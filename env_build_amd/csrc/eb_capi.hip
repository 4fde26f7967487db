// eb_capi.hip — the C-ABI of include/envbuild.h on top of the HIP kernels (libenvbuild_hip.so).
//
// A handle owns: its device ordinal, a stream, the device copies of the reference-path tables
// (full resolution + the stride-10 (x,y) table the closest-point search scans) and the per-slot
// vehicle-mode table.  Every data pointer passed to the compute entry points is a DEVICE pointer
// owned by the caller; nothing here allocates per call, synchronises, or falls back to the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/envbuild.h"
#include "eb_kernels.h"

namespace {

thread_local char g_err[512];

int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
int fail_hip(const char* what, hipError_t e) {
    std::snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return EB_EDEVICE;
}
#define EB_HIP(call)                                      \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return fail_hip(#call, e_); \
    } while (0)

}  // namespace

struct eb_handle_s {
    eb_config cfg;
    eb::PathTables pt;        // device pointers
    float* d_tables;          // one allocation: x|y|phi per path, then the stride-10 (x,y) tables
    float2* d_red_all;
    float* d_rad_all;         // 3 x 32 block radii for the pruned closest-point search
    float* d_phi10_all;       // stride-10 headings, indexed like d_red_all
    uint32_t* d_cells;        // closest-point cell grid (PathTables::cells)
    void* grids_ref;          // std::shared_ptr<const GridSet>*: the host copy of the levels these tables use (eb_debug_check_grids)
    double* d_partials;       // SUMMARY_MAX_PARTS x 6 doubles: stage-1 partials of eb_episode_summary
    eb::PathTables* d_pt;     // device copy of pt (+ slot turns) read by the rollout kernel
    int n_cu;                 // compute units of the device (persistent grid size)
    int red_off[3];
    int red_total;
    eb::VehModes modes;
    int modes_set;
    long long* trace;         // profiling aid, see eb_debug_set_trace
    long long trace_words;    //   its capacity in 64-bit words
    int stage_paths;          // -1 = by grid size; 0 / 1: the tape / gated kernels' LDS copy of the path tables off / on (eb_debug_set_stage_paths)
    int env_waves;            // 0 = by grid size; 4 / 8: waves per block of the one-launch env step (eb_debug_set_env_waves)
    int scan_one_trip;        // 1: the closest-point range scan one group of table entries per loop trip, as rounds 1-4 (eb_debug_set_scan_prefetch 0)
    eb::ExitConsts xc;        // cos / sin of the exit angles (eb_get_obs with exit ids, eb_exit_frame)
    hipStream_t gate_stream;  // producer stream of gated rollouts (high priority: a hardware queue of its own), made on first use
    hipEvent_t gate_event;    // orders eb_gate_feed behind the caller's stream (after_stream)
    int tile_variant;         // -1 = pick by batch size; 0..2 force a tile shape (eb_debug_set_tile)
    int tape_stepwise;        // 1: eb_rollout_tape runs H per-step launches instead of the tape kernel (eb_debug_set_tape_stepwise)
    int sched_rolling, sched_progress;   // -1 = by grid size; 0 / 1: the per-step kernel's rolling record loads / issue priority by progress (eb_debug_set_rollout_sched)
    // the accumulating rollouts in flight on this handle: what the step-0 launch of a workspace ran with — the later steps and the fold
    // must see the same grid, batch and horizon (a tile shape forced in between would shift every record: refused, not folded)
    struct AccRun { const void* ws; int grid, n_env, horizon; } acc_runs[4];
    int acc_next;
};

// the kernel-visible copy of the tables: refreshed whenever paths or slot modes change
static hipError_t upload_tables(eb_handle_s* h);

static int obs_dim(const eb_config& c) { return 6 + 3 * (c.n_future + 1) + 4 * c.n_veh; }
static hipStream_t pick(eb_handle, void* stream) { return (hipStream_t)stream; }   // NULL = the HIP null stream

// Closest-point cell grid.  For every 0.5 m cell C of a grid around the paths and every path k, the
// index range [lo, hi] of the stride-10 table outside of which no point can be the closest one for
// any position p in C:  with c the cell centre, hd its half diagonal and m = min_r |c - P_r|,
//   D*(p) <= |p - P_rmin| <= m + hd   and   |p - P_r| >= |c - P_r| - hd,
// so every r with |c - P_r| > m + 2 hd is strictly farther than the winner; [lo, hi] is the envelope of
// the others, widened by 0.01 m of slack — three orders of magnitude above the fp32 rounding of the
// kernel's cell assignment and of its dist^2 values at these coordinates (<= 1e-5 m).  The kernel
// scans [lo, hi] in index order with the reference's fp32 expression and a strict '<', which is the
// reference's full-scan argmin (first minimum) restricted to a range that provably contains it.
struct CellGrid {
    std::vector<uint32_t> cells;
    double x0, y0;
    int nx, ny;
};
// cell: edge length (m); margin: how far the grid reaches beyond the paths' bounding box; max_range: a cell whose range would be longer
// is marked 0xffffffff (the pruned full search is the cheaper exact answer there)
// tight: narrow the range by witnesses.  The radius bound above admits every entry within m + 2h of the cell centre — abreast of a
// straight at distance m that is +-sqrt(4 m h) entries, 76 of them for an 8 m cell 60 m out, although the closest entry of any one
// position in the cell lies within a few metres of the centre's.  Entry r can be the (fp32) first minimum at a position p of the cell
// only if it is no farther from p than any other entry w, up to the rounding of the two squared distances; |p - P_w|^2 - |p - P_r|^2
// is LINEAR in p, so its maximum over the (slightly enlarged) cell is attained at a corner: r stays only if, for each of five witnesses
// w — the closest entries of the centre and of the four corners — some corner has |p - P_w|^2 - |p - P_r|^2 >= -eps, with eps above
// the fp32 rounding of those squares at this range (2e-6 of the largest squared distance in play; the kernel's expression has a
// relative error of ~2.4e-7 per square).  The range becomes [first survivor, last survivor]: still a superset of every position's first minimum.
// two_ranges (the coarse levels): the cell word holds one or two index ranges of at most 64 entries each (coarse_cell_ranges, eb_device.h)
static int build_cell_grid(const int* red_off, const int* red_len, const float* hred, int n_paths, double cell, double margin, int max_range,
                           bool tight, bool two_ranges, CellGrid* out) {
    double x0 = 1e30, x1 = -1e30, y0 = 1e30, y1 = -1e30;
    for (int k = 0; k < n_paths; ++k) {
        const float* r = hred + 2 * (size_t)red_off[k];
        for (int i = 0; i < red_len[k]; ++i) {
            if (!std::isfinite(r[2 * i]) || !std::isfinite(r[2 * i + 1])) return fail(EB_EINVAL, "eb_set_paths: non-finite path point");
            x0 = std::min(x0, (double)r[2 * i]); x1 = std::max(x1, (double)r[2 * i]);
            y0 = std::min(y0, (double)r[2 * i + 1]); y1 = std::max(y1, (double)r[2 * i + 1]);
        }
    }
    x0 = std::floor(x0 - margin); y0 = std::floor(y0 - margin);
    int nx = (int)std::ceil((x1 + margin - x0) / cell), ny = (int)std::ceil((y1 + margin - y0) / cell);
    nx = std::min(nx, 1024); ny = std::min(ny, 1024);   // positions outside the grid take the next level / the pruned full search
    const double hd = cell * std::sqrt(2.0) / 2.0, win = 2.0 * hd + 0.01;
    out->cells.assign((size_t)n_paths * nx * ny, 0u);
    std::vector<double> d;
    for (int k = 0; k < n_paths; ++k) {
        const float* r = hred + 2 * (size_t)red_off[k];
        const int n = red_len[k];
        d.resize(n);
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix) {
                const double cx = x0 + (ix + 0.5) * cell, cy = y0 + (iy + 0.5) * cell;
                double m = 1e300;
                for (int i = 0; i < n; ++i) {
                    const double dx = cx - (double)r[2 * i], dy = cy - (double)r[2 * i + 1];
                    d[i] = dx * dx + dy * dy;
                    m = std::min(m, d[i]);
                }
                const double lim = (std::sqrt(m) + win) * (std::sqrt(m) + win);
                int lo = 0, hi = n - 1;
                while (d[lo] > lim) ++lo;
                while (d[hi] > lim) --hi;
                if (tight && hi - lo + 1 > 4) {
                    const double hc = cell / 2.0 + 0.01;                               // (a position the kernel's fp32 floor puts in this cell is well inside the enlarged one)
                    const double px[4] = {cx - hc, cx + hc, cx - hc, cx + hc}, py[4] = {cy - hc, cy - hc, cy + hc, cy + hc};
                    int wit[5];
                    wit[0] = (int)(std::min_element(d.begin() + lo, d.begin() + hi + 1) - d.begin());
                    double dk[4][512];
                    for (int k = 0; k < 4; ++k) {
                        double best = 1e300;
                        wit[k + 1] = lo;
                        for (int i = lo; i <= hi; ++i) {
                            const double dx = px[k] - (double)r[2 * i], dy = py[k] - (double)r[2 * i + 1];
                            dk[k][i] = dx * dx + dy * dy;
                            if (dk[k][i] < best) { best = dk[k][i]; wit[k + 1] = i; }
                        }
                    }
                    const double far = std::sqrt(m) + 3.0 * hd + 1.0, eps = 2e-6 * far * far + 1e-3;   // (every distance in play is below `far`; the kernel's two squares differ from the exact ones by < 4.8e-7 far^2 together)
                    auto survives = [&](int i) {
                        for (int w = 0; w < 5; ++w) {
                            double g = -1e300;
                            for (int k = 0; k < 4; ++k) g = std::max(g, dk[k][wit[w]] - dk[k][i]);
                            if (g < -eps) return false;
                        }
                        return true;
                    };
                    while (lo < hi && !survives(lo)) ++lo;
                    while (hi > lo && !survives(hi)) --hi;
                    if (two_ranges && hi - lo + 1 > max_range) {
                        // still long: a cell on the medial axis of the path (as close to one stretch as to another, e.g. the two legs
                        // of a turn) — the survivors are two clusters with the unreachable stretch between them.  Cut at the widest gap.
                        int g_lo = -1, g_hi = -1, prev = lo;
                        for (int i = lo + 1; i <= hi; ++i) {
                            if (!survives(i)) continue;
                            if (i - prev > g_hi - g_lo) { g_lo = prev; g_hi = i; }
                            prev = i;
                        }
                        if (g_lo >= 0 && g_lo - lo + 1 <= 8 && hi - g_hi + 1 <= 8) {   // (a scan trip is four entries and ~1 us in a loaded step kernel: two trips per cluster at most)
                            out->cells[((size_t)k * ny + iy) * nx + ix] = (uint32_t)lo | (uint32_t)(g_lo - lo) << 9 | (uint32_t)g_hi << 15 |
                                                                         (uint32_t)(hi - g_hi) << 24 | 1u << 30;
                            continue;
                        }
                    }
                }
                if (two_ranges)   // [lo, lo + len): lo in bits 0-8, len - 1 in 9-14; a second range in 15-23 / 24-29 when bit 30 is set;
                    // bit 31: a long range [lo, hi] (hi in bits 9-17) — the pruned search, over the blocks of that range only
                    out->cells[((size_t)k * ny + iy) * nx + ix] = hi - lo + 1 > std::min(max_range, 64) ? (0x80000000u | (uint32_t)lo | (uint32_t)hi << 9)
                                                                                                            : ((uint32_t)lo | (uint32_t)(hi - lo) << 9);
                else
                out->cells[((size_t)k * ny + iy) * nx + ix] = hi - lo + 1 > max_range ? 0xffffffffu : ((uint32_t)lo | ((uint32_t)hi << 16));
            }
    }
    out->x0 = x0; out->y0 = y0; out->nx = nx; out->ny = ny;
    return EB_OK;
}

// The four levels of one set of path tables: every level's cell words back to back (one device buffer), the levels' geometry.
static const double GRID_LVL_CELL[3] = {4.0, 32.0, 256.0}, GRID_LVL_MARGIN[3] = {250.0, 2000.0, 16000.0};
struct GridSet {
    std::vector<uint32_t> cells;    // [fine | 4 m | 32 m | 256 m]
    CellGrid fine, lvl[3];          // geometry only (their own `cells` are moved into the buffer above)
    size_t lvl_off[3];
    std::vector<float> key;         // the stride-10 tables the set was built from
    std::vector<int> key_len;
};
// eb_set_paths builds the levels on the host (~0.3 s for the three native paths); every handle of a task names the same tables, and a
// process makes many handles (one per model / env / traffic handle): the sets are kept, keyed by the tables' bytes.
static int grid_set_for(const int* red_off, const int* red_len, const float* hred, int n_paths, std::shared_ptr<const GridSet>* out) {
    static std::mutex mu;
    static std::vector<std::shared_ptr<const GridSet>> kept;
    const size_t n_float = 2 * (size_t)(red_off[n_paths - 1] + red_len[n_paths - 1]);
    std::vector<int> lens(red_len, red_len + n_paths);
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const auto& g : kept)
            if (g->key_len == lens && g->key.size() == n_float && std::memcmp(g->key.data(), hred, n_float * sizeof(float)) == 0) {
                *out = g;
                return EB_OK;
            }
    }
    auto g = std::make_shared<GridSet>();
    // (a corridor cell whose narrowed range is still long sits on the path's medial axis — inside the junction, as close to one leg of a turn
    // as to the other: its survivors are two clusters with the unreachable stretch between them, 100+ entries as ONE range, 25 scan
    // trips for an ego that wanders there.  Such cells are marked 0xffffffff = "not on this level": the 4 m level names the two clusters.)
    int rc = build_cell_grid(red_off, red_len, hred, n_paths, 1.0 / (double)eb::CELL_INV, 20.0, 16, true, false, &g->fine);
    if (rc) return rc;
    g->cells.swap(g->fine.cells);
    for (int l = 0; l < 3; ++l) {
        rc = build_cell_grid(red_off, red_len, hred, n_paths, GRID_LVL_CELL[l], GRID_LVL_MARGIN[l], 16, true, true, &g->lvl[l]);   // (ranges of up to 16 entries are scanned — four trips; longer ones go to the block search of the range)
        if (rc) return rc;
        g->lvl_off[l] = g->cells.size();
        g->cells.insert(g->cells.end(), g->lvl[l].cells.begin(), g->lvl[l].cells.end());
        std::vector<uint32_t>().swap(g->lvl[l].cells);
    }
    g->key.assign(hred, hred + n_float);
    g->key_len = lens;
    std::lock_guard<std::mutex> lock(mu);
    if (kept.size() >= 16) kept.erase(kept.begin());   // (tables set by hand, one after another: the oldest set goes)
    kept.push_back(g);
    *out = g;
    return EB_OK;
}

static hipError_t upload_tables(eb_handle_s* h) {
    for (int k = 0; k < 3; ++k) h->pt.red_off[k] = h->red_off[k];
    std::memcpy(h->pt.turn, h->modes.turn, sizeof h->pt.turn);
    return hipMemcpy(h->d_pt, &h->pt, sizeof h->pt, hipMemcpyHostToDevice);
}

extern "C" {

const char* eb_last_error(void) { return g_err; }
int eb_abi_version(void) { return EB_ABI_VERSION; }
const char* eb_backend(void) { return "hip"; }

int eb_create(const eb_config* cfg, eb_handle* out) {
    if (!cfg || !out) return fail(EB_EINVAL, "eb_create: null argument");
    if (cfg->abi_version != EB_ABI_VERSION) return fail(EB_EINVAL, "eb_create: ABI version mismatch");
    if (cfg->task < 0 || cfg->task > 2) return fail(EB_EINVAL, "eb_create: task must be left/straight/right");
    if (cfg->n_veh < 1 || cfg->n_veh > EB_MAX_VEH) return fail(EB_EINVAL, "eb_create: n_veh out of range");
    if (cfg->n_future < 0 || cfg->n_future > 64) return fail(EB_EINVAL, "eb_create: n_future out of range");
    if (cfg->mode != EB_MODE_TRAINING && cfg->mode != EB_MODE_SELECTING) return fail(EB_EINVAL, "eb_create: bad mode");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev < 1)
        return fail(EB_EDEVICE, "eb_create: no HIP device visible (libenvbuild_hip has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= n_dev) return fail(EB_EINVAL, "eb_create: device ordinal out of range");
    EB_HIP(hipSetDevice(cfg->device));
    eb_handle h = new (std::nothrow) eb_handle_s();
    if (!h) return fail(EB_ENOMEM, "eb_create: out of memory");
    std::memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    h->tile_variant = -1;
    h->stage_paths = -1;
    h->sched_rolling = -1; h->sched_progress = -1;
    {   // rotate_coordination's coordi_rotate_d * math.pi / 180 and math.cos / math.sin (UTL:130-132), by the host's libm
        const int ang[4] = {0, 90, 180, -90};   // multi_ego.py:33
        for (int k = 0; k < 4; ++k) {
            const double r = ang[k] * 3.141592653589793 / 180, ri = -ang[k] * 3.141592653589793 / 180;
            h->xc.c[k] = std::cos(r); h->xc.s[k] = std::sin(r);
            h->xc.cf[k] = (float)std::cos(r); h->xc.sf[k] = (float)std::sin(r);
            h->xc.cf[4 + k] = (float)std::cos(ri); h->xc.sf[4 + k] = (float)std::sin(ri);
        }
    }
    {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, cfg->device);
        if (e != hipSuccess) { delete h; return fail_hip("hipGetDeviceProperties", e); }
        h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    e = hipMalloc(reinterpret_cast<void**>(&h->d_partials), sizeof(double) * 6 * eb::SUMMARY_MAX_PARTS);
    if (e != hipSuccess) { delete h; return fail_hip("hipMalloc(summary partials)", e); }
    e = hipMalloc(reinterpret_cast<void**>(&h->d_pt), sizeof(eb::PathTables));
    if (e != hipSuccess) { (void)hipFree(h->d_partials); delete h; return fail_hip("hipMalloc(tables)", e); }
    *out = h;
    return EB_OK;
}

int eb_destroy(eb_handle h) {
    if (!h) return EB_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    if (h->d_tables) (void)hipFree(h->d_tables);
    if (h->d_cells) (void)hipFree(h->d_cells);
    delete static_cast<std::shared_ptr<const GridSet>*>(h->grids_ref);
    if (h->d_partials) (void)hipFree(h->d_partials);
    if (h->d_pt) (void)hipFree(h->d_pt);
    if (h->gate_stream) (void)hipStreamDestroy(h->gate_stream);
    if (h->gate_event) (void)hipEventDestroy(h->gate_event);
    delete h;
    return EB_OK;
}

int eb_sync(eb_handle h) {
    if (!h) return fail(EB_EINVAL, "eb_sync: null handle");
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(hipDeviceSynchronize());
    return EB_OK;
}

int eb_set_paths(eb_handle h, const float* xs, const float* ys, const float* phis, const int32_t* lens,
                 int32_t n_paths) {
    if (!h || !xs || !ys || !phis || !lens) return fail(EB_EINVAL, "eb_set_paths: null argument");
    if (n_paths < 1 || n_paths > EB_MAX_PATHS) return fail(EB_EINVAL, "eb_set_paths: n_paths out of range");
    size_t total = 0, red_total = 0;
    for (int k = 0; k < n_paths; ++k) {
        if (lens[k] < 3) return fail(EB_EINVAL, "eb_set_paths: path too short");
        if ((lens[k] + 9) / 10 > 512) return fail(EB_EINVAL, "eb_set_paths: path longer than 5120 points (32 search blocks)");
        total += (size_t)lens[k];
        red_total += (size_t)(lens[k] + 9) / 10;   // len(np.arange(0, path_len, 10)), DAM:704
    }
    for (size_t i = 0; i < total; ++i)
        if (!std::isfinite(xs[i]) || !std::isfinite(ys[i]) || !std::isfinite(phis[i]))
            return fail(EB_EINVAL, "eb_set_paths: non-finite path point");
    // ---- everything below is built into temporaries; the handle changes only once all of it has succeeded ----
    // host staging: [x | y | phi] full resolution, then float2 stride-10 tables
    std::vector<float> host(3 * total + 2 * red_total + 4 + 96 + 8 + red_total + 8 + 8, 0.0f);
    float* hx = host.data();
    float* hy = hx + total;
    float* hp = hy + total;
    size_t red_byte_off = (3 * total * sizeof(float) + 15) / 16 * 16;   // 16-byte aligned float2 table
    float* hred = reinterpret_cast<float*>(reinterpret_cast<char*>(host.data()) + red_byte_off);
    std::memcpy(hx, xs, total * sizeof(float));
    std::memcpy(hy, ys, total * sizeof(float));
    std::memcpy(hp, phis, total * sizeof(float));
    int red_off[3] = {0, 0, 0}, red_len[3] = {0, 0, 0};
    size_t off = 0, roff = 0;
    for (int k = 0; k < n_paths; ++k) {
        red_off[k] = (int)roff;
        red_len[k] = (lens[k] + 9) / 10;
        for (int i = 0; i < lens[k]; i += 10) {
            hred[2 * roff] = xs[off + i];
            hred[2 * roff + 1] = ys[off + i];
            ++roff;
        }
        off += (size_t)lens[k];
    }
    // block radii of the pruned search: R_b >= max_r |P_r - c_b| over block b = [16b, 16b+16), with
    // c_b = P_min(16b+8, n-1); evaluated in double on the fp32 table values and inflated by 1e-4 m
    const size_t rad_byte_off = (red_byte_off + 2 * red_total * sizeof(float) + 15) / 16 * 16;
    float* hrad = reinterpret_cast<float*>(reinterpret_cast<char*>(host.data()) + rad_byte_off);
    for (int i = 0; i < 96; ++i) hrad[i] = 0.0f;
    for (int k = 0; k < n_paths; ++k) {
        const float* rx = hred + 2 * (size_t)red_off[k];
        const int n = red_len[k];
        for (int b = 0; 16 * b < n; ++b) {
            const int c = std::min(16 * b + 8, n - 1);
            double r2 = 0.0;
            for (int r = 16 * b; r < std::min(16 * b + 16, n); ++r) {
                const double dx = (double)rx[2 * r] - (double)rx[2 * c], dy = (double)rx[2 * r + 1] - (double)rx[2 * c + 1];
                r2 = std::max(r2, dx * dx + dy * dy);
            }
            hrad[32 * k + b] = (float)(std::sqrt(r2) * (1.0 + 1e-6) + 1e-4);
        }
    }
    // stride-10 headings (phi of the point the search returns), same indexing as the (x, y) table; both
    // tables are readable 4 entries past their end (the range scan loads 4 points at a time)
    const size_t phi10_byte_off = rad_byte_off + 96 * sizeof(float);
    float* hphi10 = reinterpret_cast<float*>(reinterpret_cast<char*>(host.data()) + phi10_byte_off);
    off = 0; roff = 0;
    for (int k = 0; k < n_paths; ++k) {
        for (int i = 0; i < lens[k]; i += 10) hphi10[roff++] = phis[off + i];
        off += (size_t)lens[k];
    }
    const size_t bytes = phi10_byte_off + (red_total + 8) * sizeof(float);
    // the corridor's level: 0.5 m cells out to 20 m around the paths (2-4 table entries per cell once narrowed by witnesses: one group
    // of the scan); three coarser ones for the egos that have left the road or finished and drive on — 4 m cells out to 250 m, 32 m out
    // to 2 km, 256 m out to 16 km; where a coarse cell's range(s) would still be long (abreast of a long straight, far out) the pruned
    // full search stays the answer, as beyond 16 km.  Built once per distinct set of tables and process: handles of one task share them.
    std::shared_ptr<const GridSet> gs;
    int rc = grid_set_for(red_off, red_len, hred, n_paths, &gs);
    if (rc) return rc;
    const CellGrid& grid = gs->fine;
    const CellGrid* lvl = gs->lvl;
    const size_t* lvl_off = gs->lvl_off;
    EB_HIP(hipSetDevice(h->cfg.device));
    float* d_tables = nullptr;
    uint32_t* d_cells = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_tables), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_cells), gs->cells.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_tables, host.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cells, gs->cells.data(), gs->cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // nothing in flight reads the old tables any more
    if (e != hipSuccess) {
        if (d_tables) (void)hipFree(d_tables);
        if (d_cells) (void)hipFree(d_cells);
        return fail_hip("eb_set_paths: device tables", e);
    }
    // ---- swap ----
    eb::PathTables pt;
    std::memset(&pt, 0, sizeof pt);
    float2* d_red_all = reinterpret_cast<float2*>(reinterpret_cast<char*>(d_tables) + red_byte_off);
    off = 0;
    for (int k = 0; k < n_paths; ++k) {
        pt.x[k] = d_tables + off;
        pt.y[k] = d_tables + total + off;
        pt.phi[k] = d_tables + 2 * total + off;
        pt.red[k] = d_red_all + red_off[k];
        pt.phi10[k] = reinterpret_cast<float*>(reinterpret_cast<char*>(d_tables) + phi10_byte_off) + red_off[k];
        pt.len[k] = lens[k];
        pt.red_len[k] = red_len[k];
        off += (size_t)lens[k];
    }
    pt.n_paths = n_paths;
    pt.cells = d_cells;
    pt.rad = reinterpret_cast<float*>(reinterpret_cast<char*>(d_tables) + rad_byte_off);
    pt.gx0 = (float)grid.x0; pt.gy0 = (float)grid.y0;   // integers: exact in fp32
    pt.gnx = grid.nx; pt.gny = grid.ny;
    for (int l = 0; l < 3; ++l)
        pt.coarse[l] = {d_cells + lvl_off[l], (float)lvl[l].x0, (float)lvl[l].y0, (float)(1.0 / GRID_LVL_CELL[l]), lvl[l].nx, lvl[l].ny};
    const eb::PathTables old_pt = h->pt;
    float* old_tables = h->d_tables;
    uint32_t* old_cells = h->d_cells;
    int old_off[3] = {h->red_off[0], h->red_off[1], h->red_off[2]};
    h->pt = pt;
    for (int k = 0; k < 3; ++k) h->red_off[k] = red_off[k];
    e = upload_tables(h);
    if (e != hipSuccess) {   // the device copy of the table descriptor still names the old tables: keep them
        h->pt = old_pt;
        for (int k = 0; k < 3; ++k) h->red_off[k] = old_off[k];
        (void)hipFree(d_tables);
        (void)hipFree(d_cells);
        return fail_hip("eb_set_paths: upload", e);
    }
    h->d_tables = d_tables;
    h->d_cells = d_cells;
    delete static_cast<std::shared_ptr<const GridSet>*>(h->grids_ref);
    h->grids_ref = new std::shared_ptr<const GridSet>(gs);
    h->d_red_all = d_red_all;
    h->d_rad_all = reinterpret_cast<float*>(reinterpret_cast<char*>(d_tables) + rad_byte_off);
    h->d_phi10_all = reinterpret_cast<float*>(reinterpret_cast<char*>(d_tables) + phi10_byte_off);
    h->red_total = (int)red_total;
    if (old_tables) (void)hipFree(old_tables);
    if (old_cells) (void)hipFree(old_cells);
    return EB_OK;
}

int eb_set_veh_modes(eb_handle h, const uint8_t* mode_id, int32_t n) {
    if (!h || !mode_id) return fail(EB_EINVAL, "eb_set_veh_modes: null argument");
    if (n != h->cfg.n_veh) return fail(EB_EINVAL, "eb_set_veh_modes: n != n_veh");
    for (int j = 0; j < n; ++j)
        if (mode_id[j] >= EB_VMODE_COUNT) return fail(EB_EINVAL, "eb_set_veh_modes: bad mode id");
    for (int j = 0; j < n; ++j) {
        h->modes.mode[j] = mode_id[j];
        switch (mode_id[j]) {   // predict_for_a_mode, DAM:416-421
            case EB_VMODE_DL: case EB_VMODE_RD: case EB_VMODE_UR: case EB_VMODE_LU: h->modes.turn[j] = eb::TURN_LEFT; break;
            case EB_VMODE_DR: case EB_VMODE_RU: case EB_VMODE_UL: case EB_VMODE_LD: h->modes.turn[j] = eb::TURN_RIGHT; break;
            default: h->modes.turn[j] = eb::TURN_NONE; break;
        }
    }
    h->modes_set = 1;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(hipDeviceSynchronize());
    EB_HIP(upload_tables(h));
    return EB_OK;
}

static int check_paths(eb_handle h, const char* who) {
    if (!h) return fail(EB_EINVAL, who);
    if (h->pt.n_paths < 1) return fail(EB_ESTATE, "paths not set (eb_set_paths)");
    return EB_OK;
}
static int check_modes(eb_handle h) {
    if (!h->modes_set) return fail(EB_ESTATE, "vehicle modes not set (eb_set_veh_modes)");
    return EB_OK;
}
static int check_rollout(eb_handle h, int n_env, const int32_t* ref_idx, int path_id, const char* who) {
    int rc = check_paths(h, who);
    if (rc) return rc;
    rc = check_modes(h);
    if (rc) return rc;
    if (h->cfg.mode == EB_MODE_TRAINING) {
        if (!ref_idx) return fail(EB_EINVAL, "training mode needs ref_idx (EnvironmentModel.reset(obses, ref_indexes))");
    } else if (path_id < 0 || path_id >= h->pt.n_paths) return fail(EB_EINVAL, "bad path_id");
    return EB_OK;
}

int eb_f_xu(eb_handle h, int32_t n, const float* states, const float* actions, float tau, float* next_states,
            float* params, void* stream) {
    if (!h || n < 0 || !states || !actions || !next_states) return fail(EB_EINVAL, "eb_f_xu: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_f_xu(n, states, actions, tau, next_states, params, pick(h, stream)));
    return EB_OK;
}

int eb_action_transform(eb_handle h, int32_t n, const float* actions, float* scaled, void* stream) {
    if (!h || n < 0 || !actions || !scaled) return fail(EB_EINVAL, "eb_action_transform: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_action_transform(n, actions, scaled, pick(h, stream)));
    return EB_OK;
}

int eb_compute_rewards(eb_handle h, int32_t n_env, const float* obs, const float* actions, float* out5,
                       float* out_dict16, void* stream) {
    if (!h || n_env < 0 || !obs || !actions || !out5) return fail(EB_EINVAL, "eb_compute_rewards: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_rewards(h->cfg.task, n_env, obs_dim(h->cfg), h->cfg.n_future, h->cfg.n_veh, obs, actions, out5,
                              out_dict16, pick(h, stream)));
    return EB_OK;
}

// obs_in / obs_out point at fp32 rows, or at binary16 rows when storage_f16 is set
struct GateArgs {
    const uint32_t* ready;
    uint32_t* done;
    void* obs_steps;
    uint32_t* status;
    int spin;
};
// the episodic accumulator of an accumulating rollout step (eb_rollout_step_acc, plans with a summary): the workspace holds the
// per-step records [horizon][blocks][ACC_RECORD_DOUBLES], then the last launch's (sum, max) pairs [blocks][2], then a byte per env
// and 64 more (the CPU library keeps 8 doubles and its per-env flags there) — `blocks` = the grid of the SMALLEST tile shape, so any shape fits
struct AccArgs {
    void* workspace;
    int step, horizon;
    const float* prev_out5;
};
// the shield's accumulation riding on the step's launch (eb_shield_is_safe): FusedArgs::shield_*
struct ShieldArgs {
    float* punish;
    uint8_t* safe;
    int row, first, last;
};
static int envs_per_tile(eb_handle h, int variant) {
    return std::max(1, std::min(64, eb::fused_tile_records(variant) / h->cfg.n_veh));
}
static size_t acc_blocks_max(eb_handle h, int32_t n_env) {
    const int e = envs_per_tile(h, 2);
    return (size_t)((n_env + e - 1) / e);
}
static size_t acc_workspace_bytes(eb_handle h, int32_t n_env, int32_t horizon) {
    return acc_blocks_max(h, n_env) * ((size_t)horizon * eb::ACC_RECORD_DOUBLES + 2) * sizeof(double) + (size_t)n_env + 64;
}
static double* acc_records(void* ws, int step, size_t grid) {
    return reinterpret_cast<double*>(ws) + (size_t)step * grid * eb::ACC_RECORD_DOUBLES;
}
static double* acc_finals(void* ws, int horizon, size_t grid) {
    return reinterpret_cast<double*>(ws) + (size_t)horizon * grid * eb::ACC_RECORD_DOUBLES;
}
static int pick_variant(eb_handle h, int32_t n_env) {
    int variant = h->tile_variant;                                        // eb_debug_set_tile
    if (variant < 0 || variant > 2) {
        // the largest tile that still gives every CU two blocks; small batches take small tiles.  A tile holds at most 64 envs (one
        // lane of the env wave each): with 16 slots or fewer the 2048-record tile would be at most half full — its record lanes idle
        // through half their records — and the 1024-record tile is the faster one at every batch size (round 6, profiles/r6_tile_sweep2.txt:
        // 65 536 envs x 16 / 9 / 8 / 5 slots 10.3 -> 9.9 / 9.7 -> 8.6 / 8.3 -> 7.6 / 8.7 -> 7.4 us per step; x 24 slots the large tile stays ahead)
        variant = 2;
        for (int v = 0; v < 2; ++v) {
            if (v == 0 && 64 * h->cfg.n_veh <= eb::fused_tile_records(1)) continue;
            const int e = std::max(1, std::min(64, eb::fused_tile_records(v) / h->cfg.n_veh));
            if ((n_env + e - 1) / e >= 2 * h->n_cu) { variant = v; break; }
        }
    }
    return variant;
}

// Small grids (at most two blocks per CU) keep the stride-10 path tables in LDS for the whole launch: their steps are bound
// by the env wave's chain of dependent table reads, not by throughput.  Larger ones leave the LDS to occupancy.  ONE
// decision for the launch (rollout_fused) and for the residency query (gated_blocks): eb_debug_set_stage_paths forces it.
static bool stage_paths_in_lds(eb_handle h, int grid) {
    return h->stage_paths < 0 ? grid <= 2 * h->n_cu : h->stage_paths != 0;
}

// How the per-step launch spends its memory queue and its issue slots (csrc/eb_rollout.hip; measured: profiles/r6_ab3-5.txt,
// r6_sched_sweep1-3*.txt — N = 8 ... 64, fp32 and binary16 rows, 2 to 16 tiles per CU).  The record waves that are behind issue first:
// on the 2048- and 1024-record tiles always (never more than 1 % slower, up to 8 % faster: 9.85 -> 9.0 us at 65 536 envs x 16), off on
// the 256-record tile (grids of a few blocks per CU).  Rolling record loads (2048-record tile): on grids of at most three tiles per CU,
// and at any size when a tile holds at most 32 envs (64 slots: + 4-5 % at 8 tiles per CU) — with 64-env tiles they cost 3 % at four
// tiles per CU and 11 % at sixteen.  Same bits every way; eb_debug_set_rollout_sched forces either.
static void rollout_sched(eb_handle h, int variant, int grid, int envs_per_tile_, int* rolling, int* by_progress) {
    *by_progress = h->sched_progress >= 0 ? h->sched_progress : (variant <= 1 ? 1 : 0);
    *rolling = variant != 0 ? 0 : h->sched_rolling >= 0 ? h->sched_rolling : (grid <= 3 * h->n_cu || envs_per_tile_ <= 32);
}

static int rollout_fused(eb_handle h, int variant, int32_t n_env, const float* obs_in, const float* actions,
                         const int32_t* ref_idx, int32_t path_id, float* obs_out, float* out5,
                         float* scaled_actions, int actions_raw, int do_rewards, hipStream_t s, int storage_f16,
                         int tape_horizon = 0,   // > 0: `actions` is a tape [H, n_env, 2], `out5` is [H, 5, n_env], one launch
                         const GateArgs* gate = nullptr, const AccArgs* acc = nullptr, const ShieldArgs* shield = nullptr) {
    const int NV = h->cfg.n_veh;
    if (tape_horizon > 0 && !gate) variant = eb::tape_tile_variant(variant, NV, storage_f16);
    eb::FusedArgs A;
    std::memset(&A, 0, sizeof A);
    A.storage_f16 = storage_f16;
    A.obs_in = obs_in; A.actions = actions; A.ref_idx = ref_idx; A.obs_out = obs_out; A.out5 = out5;
    A.scaled_actions = scaled_actions;
    A.dt = h->d_pt;
    A.xy10 = reinterpret_cast<const float*>(h->d_red_all);
    A.phi10 = h->d_phi10_all;
    A.rad_all = h->d_rad_all;
    A.cells = h->d_cells;
    A.gx0 = h->pt.gx0; A.gy0 = h->pt.gy0; A.gnx = h->pt.gnx; A.gny = h->pt.gny;
    for (int k = 0; k < 3; ++k) { A.red_off[k] = h->red_off[k]; A.red_len[k] = h->pt.red_len[k]; }
    A.n_paths = h->pt.n_paths;
    A.n_env = n_env; A.obs_dim = obs_dim(h->cfg); A.n_veh = NV; A.n_future = h->cfg.n_future;
    A.nv_magic = NV == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)NV - 1) / (unsigned)NV);   // 0: item / 1
    A.envs_per_tile = envs_per_tile(h, variant);
    A.path_id = path_id;
    A.training = h->cfg.mode == EB_MODE_TRAINING;
    A.actions_raw = actions_raw;
    A.do_rewards = do_rewards;
    A.trace = h->trace; A.trace_words = h->trace_words; A.scan_one_trip = h->scan_one_trip;
    if (shield) {
        A.shield_punish = shield->punish; A.shield_safe = shield->safe; A.shield_row = shield->row;
        A.shield_first = shield->first; A.shield_last = shield->last;
    }
    if (gate) {
        A.gate_ready = gate->ready; A.gate_done = gate->done; A.gate_obs = gate->obs_steps; A.gate_status = gate->status;
        A.gate_spin = gate->spin;
    }
    const int grid = (n_env + A.envs_per_tile - 1) / A.envs_per_tile;
    if (tape_horizon == 0) rollout_sched(h, variant, grid, A.envs_per_tile, &A.rolling, &A.by_progress);
    if (acc) {   // records are indexed by THIS grid (the same at every step of a rollout: one handle state, one n_env)
        // the rollout a workspace belongs to is fixed by its step-0 launch
        eb_handle_s::AccRun* run = nullptr;
        for (auto& r : h->acc_runs) if (r.ws == acc->workspace) run = &r;
        if (acc->step == 0) {
            if (!run) { run = &h->acc_runs[h->acc_next]; h->acc_next = (h->acc_next + 1) % 4; }
            *run = eb_handle_s::AccRun{acc->workspace, grid, n_env, acc->horizon};
        } else if (run && (run->grid != grid || run->n_env != n_env || run->horizon != acc->horizon)) {
            return fail(EB_EINVAL, "eb_rollout_step_acc: this workspace's rollout was started with another grid, batch or horizon "
                                   "(its step-0 launch fixes them; eb_debug_set_tile in between?)");
        }
        A.acc_rec = acc_records(acc->workspace, acc->step, (size_t)grid);
        if (acc->step > 0) { A.prev_out5 = acc->prev_out5; A.prev_rec = acc_records(acc->workspace, acc->step - 1, (size_t)grid); }
        if (acc->step == acc->horizon - 1) A.acc_final = acc_finals(acc->workspace, acc->horizon, (size_t)grid);
    }
    if (tape_horizon > 0 && stage_paths_in_lds(h, grid)) A.stage_entries = h->red_total + 4;
    if (tape_horizon > 0) EB_HIP(eb::launch_rollout_tape_fused(h->cfg.task, variant, A, tape_horizon, grid, s));
    else EB_HIP(eb::launch_rollout_fused(h->cfg.task, variant, A, grid, s));
    return EB_OK;
}

static int rollout_common(eb_handle h, int32_t n_env, const float* obs_in, const float* actions,
                          const int32_t* ref_idx, int32_t path_id, float* obs_out, float* out5,
                          float* scaled_actions, int actions_raw, int do_rewards, hipStream_t s, int storage_f16 = 0,
                          int tape_horizon = 0, const GateArgs* gate = nullptr, const AccArgs* acc = nullptr,
                          const ShieldArgs* shield = nullptr) {
    return rollout_fused(h, pick_variant(h, n_env), n_env, obs_in, actions, ref_idx, path_id, obs_out, out5, scaled_actions,
                         actions_raw, do_rewards, s, storage_f16, tape_horizon, gate, acc, shield);
}

// blocks of a gated rollout over n_env envs, or 0 when they cannot all be resident at once next to a producer: a gated
// rollout may take HALF of the device's block slots — whatever opens its gates has to run beside it, and a grid that
// fills every CU leaves the producer's workgroups nowhere to go (both sides would then wait until they give up)
static int gated_blocks(eb_handle h, int32_t n_env, int* variant_out = nullptr) {
    // the tile shape the per-step kernel would take for this batch, or the next larger one whose grid fits (a forced shape stays)
    const int first = pick_variant(h, n_env);
    for (int variant = first; variant >= 0; --variant) {
        const int ept = std::max(1, std::min(64, eb::fused_tile_records(variant) / h->cfg.n_veh));
        const int grid = (n_env + ept - 1) / ept;
        const size_t dyn = stage_paths_in_lds(h, grid) ? (size_t)(h->red_total + 4) * 12 : 0;   // as rollout_fused will launch it
        const int per_cu = eb::tape_blocks_per_cu(h->cfg.task, variant, h->cfg.n_veh, 0, dyn);
        if (grid <= per_cu * h->n_cu / 2) {
            if (variant_out) *variant_out = variant;
            return grid;
        }
        if (h->tile_variant >= 0) break;
    }
    return 0;
}

int eb_compute_next_obses(eb_handle h, int32_t n_env, const float* obs, const float* actions,
                          const int32_t* ref_idx, int32_t path_id, float* obs_out, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_compute_next_obses: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs || !actions || !obs_out) return fail(EB_EINVAL, "eb_compute_next_obses: bad argument");
    if (obs == obs_out) return fail(EB_EINVAL, "eb_compute_next_obses: in-place update is not supported");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    return rollout_common(h, n_env, obs, actions, ref_idx, path_id, obs_out, nullptr, nullptr, 0, 0, pick(h, stream));
}

int eb_rollout_step(eb_handle h, int32_t n_env, const float* obs_in, const float* actions, const int32_t* ref_idx,
                    int32_t path_id, float* obs_out, float* out5, float* scaled_actions, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_step: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs_in || !actions || !obs_out || !out5) return fail(EB_EINVAL, "eb_rollout_step: bad argument");
    if (obs_in == obs_out) return fail(EB_EINVAL, "eb_rollout_step: in-place update is not supported");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    return rollout_common(h, n_env, obs_in, actions, ref_idx, path_id, obs_out, out5, scaled_actions, 1, 1,
                          pick(h, stream));
}

int eb_episode_acc_bytes(eb_handle h, int32_t n_env, int32_t horizon, int64_t* bytes) {
    if (!h || n_env < 0 || horizon < 0 || !bytes) return fail(EB_EINVAL, "eb_episode_acc_bytes: bad argument");
    *bytes = (int64_t)acc_workspace_bytes(h, n_env, horizon);
    return EB_OK;
}

int eb_rollout_step_acc(eb_handle h, int32_t n_env, const float* obs_in, const float* actions, const int32_t* ref_idx,
                        int32_t path_id, float* obs_out, float* out5, float* scaled_actions, void* acc, int32_t step,
                        int32_t horizon, const float* prev_out5, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_step_acc: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs_in || !actions || !obs_out || !out5 || !acc || horizon < 1 || step < 0 || step >= horizon ||
        (step > 0 && !prev_out5) || prev_out5 == out5)
        return fail(EB_EINVAL, "eb_rollout_step_acc: bad argument (0 <= step < horizon; prev_out5 = the previous step's out5 for step > 0)");
    if (((uintptr_t)acc & 15) != 0) return fail(EB_EINVAL, "eb_rollout_step_acc: acc must be 16-byte aligned");
    if (obs_in == obs_out) return fail(EB_EINVAL, "eb_rollout_step_acc: in-place update is not supported");
    EB_HIP(hipSetDevice(h->cfg.device));
    const AccArgs a{acc, step, horizon, prev_out5};
    return rollout_common(h, n_env, obs_in, actions, ref_idx, path_id, obs_out, out5, scaled_actions, 1, 1, pick(h, stream), 0, 0,
                          nullptr, &a);
}

int eb_episode_acc_finish(eb_handle h, int32_t n_env, int32_t horizon, const void* acc, float* out8, void* stream) {
    if (!h || n_env < 0 || horizon < 1 || !out8 || (n_env > 0 && !acc)) return fail(EB_EINVAL, "eb_episode_acc_finish: bad argument");
    EB_HIP(hipSetDevice(h->cfg.device));
    const int e = envs_per_tile(h, pick_variant(h, n_env));
    size_t grid = (size_t)((n_env + e - 1) / e);
    // the grid the accumulating launches ran on: the one their step-0 launch recorded (a workspace this handle has not seen a step 0
    // of — filled through another handle of the same shape — is folded with the handle's current grid)
    for (const auto& r : h->acc_runs)
        if (r.ws == acc && r.ws) {
            if (r.n_env != n_env || r.horizon != horizon)
                return fail(EB_EINVAL, "eb_episode_acc_finish: n_env / horizon differ from the rollout that filled this workspace");
            grid = (size_t)r.grid;
        }
    void* ws = const_cast<void*>(acc);
    EB_HIP(eb::launch_acc_fold((int)grid, n_env, horizon, acc_records(ws, 0, grid), acc_finals(ws, horizon, grid), out8,
                               pick(h, stream)));
    return EB_OK;
}

// H launches of the per-step kernel, ping-ponging so that the last step lands in obs_out (what eb_plan_* records);
// acc != NULL: accumulating launches (the episodic summary's sums collected on the way)
static int rollout_tape_stepwise(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                                 const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out,
                                 float* out5_steps, hipStream_t s, int storage_f16, void* acc = nullptr) {
    const float* cur = obs_in;
    for (int t = 0; t < horizon; ++t) {
        float* dst = ((horizon - 1 - t) % 2 == 0) ? obs_out : obs_work;
        const AccArgs a{acc, t, horizon, t > 0 ? out5_steps + (size_t)(t - 1) * 5 * n_env : nullptr};
        int rc = rollout_common(h, n_env, cur, action_tape + (size_t)t * n_env * 2, ref_idx, path_id, dst,
                                out5_steps + (size_t)t * 5 * n_env, nullptr, 1, 1, s, storage_f16, 0, nullptr, acc ? &a : nullptr);
        if (rc) return rc;
        cur = dst;
    }
    return EB_OK;
}

// Open loop over a tape: ONE launch of the tape kernel (state in registers across the steps); bit-identical to the
// H per-step launches (eb_debug_set_tape_stepwise forces those, for A/B checks).
static int rollout_tape_any(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                            const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out, float* out5_steps,
                            hipStream_t s, int storage_f16) {
    if (h->tape_stepwise)
        return rollout_tape_stepwise(h, n_env, horizon, obs_in, action_tape, ref_idx, path_id, obs_work, obs_out, out5_steps, s,
                                     storage_f16);
    return rollout_common(h, n_env, obs_in, action_tape, ref_idx, path_id, obs_out, out5_steps, nullptr, 1, 1, s, storage_f16,
                          horizon);
}

int eb_rollout_tape(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                    const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out, float* out5_steps,
                    void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_tape: null handle");
    if (rc) return rc;
    if (n_env < 0 || horizon < 1 || !obs_in || !action_tape || !obs_work || !obs_out || !out5_steps)
        return fail(EB_EINVAL, "eb_rollout_tape: bad argument");
    if (obs_work == obs_out || obs_in == obs_work || obs_in == obs_out)
        return fail(EB_EINVAL, "eb_rollout_tape: obs_in, obs_work and obs_out must be distinct buffers");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    return rollout_tape_any(h, n_env, horizon, obs_in, action_tape, ref_idx, path_id, obs_work, obs_out, out5_steps,
                            pick(h, stream), 0);
}

int eb_rollout_gated_blocks(eb_handle h, int32_t n_env, int32_t* n_blocks) {
    if (!h || n_env < 0 || !n_blocks) return fail(EB_EINVAL, "eb_rollout_gated_blocks: bad argument");
    // the answer depends on the staged table size and the slot count: the handle must be configured as for the launch
    int rc = check_paths(h, "eb_rollout_gated_blocks: null handle");
    if (!rc) rc = check_modes(h);
    if (rc) return rc;
    EB_HIP(hipSetDevice(h->cfg.device));
    *n_blocks = n_env == 0 ? 0 : gated_blocks(h, n_env);
    return EB_OK;
}

int eb_rollout_gated(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                     const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out, float* out5_steps,
                     float* obs_steps, const uint32_t* step_ready, uint32_t* step_done, int32_t n_blocks,
                     uint32_t* status, int32_t spin_limit, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_gated: null handle");
    if (rc) return rc;
    if (n_env < 0 || horizon < 1 || !obs_in || !action_tape || !obs_work || !obs_out || !out5_steps || !step_ready || !step_done ||
        !status || spin_limit < 1)
        return fail(EB_EINVAL, "eb_rollout_gated: bad argument");
    if (obs_work == obs_out || obs_in == obs_work || obs_in == obs_out)
        return fail(EB_EINVAL, "eb_rollout_gated: obs_in, obs_work and obs_out must be distinct buffers");
    EB_HIP(hipSetDevice(h->cfg.device));
    int variant = 0;
    const int grid = gated_blocks(h, n_env, &variant);
    if (grid == 0)
        return fail(EB_EINVAL, "eb_rollout_gated: n_env needs more than half of the device's block slots (a gated rollout must be fully resident, next to its producer)");
    // step_done was sized by the caller from an earlier eb_rollout_gated_blocks: the grid about to be launched (tile
    // shape, staging, occupancy — all re-derived from the handle's present state) has to be THAT grid, or the done
    // records would be written past the buffer and the consumer would count the wrong number of them
    if (n_blocks != grid)
        return fail(EB_EINVAL, "eb_rollout_gated: n_blocks does not match the grid this handle launches now (call eb_rollout_gated_blocks again)");
    const GateArgs g{step_ready, step_done, obs_steps, status, spin_limit};
    return rollout_fused(h, variant, n_env, obs_in, action_tape, ref_idx, path_id, obs_out, out5_steps, nullptr, 1, 1, pick(h, stream), 0,
                         horizon, &g);
}

int eb_gate_feed(eb_handle h, int32_t n_env, int32_t horizon, int32_t n_blocks, const float* staged_tape,
                 float* live_tape, uint32_t* step_ready, const uint32_t* step_done, uint32_t* status,
                 int32_t spin_limit, void* after_stream, int32_t wait_after, void* stream) {
    if (!h || n_env < 1 || (n_env & 1) || horizon < 1 || n_blocks < 1 || !staged_tape || !live_tape || !step_ready || !step_done ||
        !status || spin_limit < 1 || staged_tape == live_tape)
        return fail(EB_EINVAL, "eb_gate_feed: bad argument (n_env even: a step's actions are copied 16 bytes at a time)");
    EB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        // The producer has to run SIDE BY SIDE with the gated rollout.  Two streams of the same priority may share a
        // hardware queue (then the second kernel would wait for the first to end, and each waits for the other's flags
        // until both give up); a high-priority stream sits in a queue of its own.
        if (!h->gate_stream) {
            int lo = 0, hi = 0;
            EB_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            EB_HIP(hipStreamCreateWithPriority(&h->gate_stream, hipStreamNonBlocking, hi));
        }
        s = h->gate_stream;
    }
    if (wait_after && (hipStream_t)after_stream != s) {
        // the feed reads staged_tape and the zeroed flags: whatever the caller enqueued on `after_stream` to produce them
        // (asynchronous fills included) must be complete first — an event, not a host wait
        if (!h->gate_event) EB_HIP(hipEventCreateWithFlags(&h->gate_event, hipEventDisableTiming));
        EB_HIP(hipEventRecord(h->gate_event, (hipStream_t)after_stream));
        EB_HIP(hipStreamWaitEvent(s, h->gate_event, 0));
    }
    EB_HIP(eb::launch_gate_feed(horizon, n_blocks, (size_t)n_env * 8, staged_tape, live_tape, step_ready, step_done, status,
                                spin_limit, s));
    return EB_OK;
}

int eb_rollout_step_f16(eb_handle h, int32_t n_env, const uint16_t* obs_in, const float* actions, const int32_t* ref_idx,
                        int32_t path_id, uint16_t* obs_out, float* out5, float* scaled_actions, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_step_f16: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs_in || !actions || !obs_out || !out5) return fail(EB_EINVAL, "eb_rollout_step_f16: bad argument");
    if (obs_in == obs_out) return fail(EB_EINVAL, "eb_rollout_step_f16: in-place update is not supported");
    EB_HIP(hipSetDevice(h->cfg.device));
    return rollout_common(h, n_env, reinterpret_cast<const float*>(obs_in), actions, ref_idx, path_id,
                          reinterpret_cast<float*>(obs_out), out5, scaled_actions, 1, 1, pick(h, stream), 1);
}

int eb_rollout_tape_f16(eb_handle h, int32_t n_env, int32_t horizon, const uint16_t* obs_in, const float* action_tape,
                        const int32_t* ref_idx, int32_t path_id, uint16_t* obs_work, uint16_t* obs_out, float* out5_steps,
                        void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_tape_f16: null handle");
    if (rc) return rc;
    if (n_env < 0 || horizon < 1 || !obs_in || !action_tape || !obs_work || !obs_out || !out5_steps)
        return fail(EB_EINVAL, "eb_rollout_tape_f16: bad argument");
    if (obs_work == obs_out || obs_in == obs_work || obs_in == obs_out)
        return fail(EB_EINVAL, "eb_rollout_tape_f16: obs_in, obs_work and obs_out must be distinct buffers");
    EB_HIP(hipSetDevice(h->cfg.device));
    return rollout_tape_any(h, n_env, horizon, reinterpret_cast<const float*>(obs_in), action_tape, ref_idx, path_id,
                            reinterpret_cast<float*>(obs_work), reinterpret_cast<float*>(obs_out), out5_steps,
                            pick(h, stream), 1);
}

int eb_find_closest_point(eb_handle h, int32_t n, const float* xs, const float* ys, const int32_t* ref_idx,
                          int32_t path_id, int32_t ratio, int32_t* out_index, float* out_points, void* stream) {
    int rc = check_paths(h, "eb_find_closest_point: null handle");
    if (rc) return rc;
    if (n < 0 || !xs || !ys || !out_index || ratio < 1) return fail(EB_EINVAL, "eb_find_closest_point: bad argument");
    if (!ref_idx && (path_id < 0 || path_id >= h->pt.n_paths)) return fail(EB_EINVAL, "eb_find_closest_point: bad path_id");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_tracking(h->cfg.task, n, h->pt, xs, ys, nullptr, nullptr, ref_idx, path_id, 0, ratio, nullptr,
                               out_index, out_points, pick(h, stream)));
    return EB_OK;
}

int eb_path_points(eb_handle h, int32_t n, const int32_t* index, const int32_t* ref_idx, int32_t path_id,
                   int32_t n_future, float* out_points, void* stream) {
    int rc = check_paths(h, "eb_path_points: null handle");
    if (rc) return rc;
    if (n < 0 || n_future < 0 || (n > 0 && (!index || !out_points))) return fail(EB_EINVAL, "eb_path_points: bad argument");
    if (!ref_idx && (path_id < 0 || path_id >= h->pt.n_paths)) return fail(EB_EINVAL, "eb_path_points: bad path_id");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_path_points(n, h->pt, index, ref_idx, path_id, n_future, out_points, pick(h, stream)));
    return EB_OK;
}

int eb_phi_diff(eb_handle h, int32_t n, const float* phi_diff, float* out, void* stream) {
    if (!h || n < 0 || (n > 0 && (!phi_diff || !out))) return fail(EB_EINVAL, "eb_phi_diff: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_phi_diff(n, phi_diff, out, pick(h, stream)));
    return EB_OK;
}

int eb_ego_predict(eb_handle h, int32_t n, const float* ego, const float* actions, float* next_ego, void* stream) {
    if (!h || n < 0 || (n > 0 && (!ego || !actions || !next_ego))) return fail(EB_EINVAL, "eb_ego_predict: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_ego_predict(n, ego, actions, next_ego, pick(h, stream)));
    return EB_OK;
}

int eb_tracking_error(eb_handle h, int32_t n, const float* xs, const float* ys, const float* phis, const float* vs,
                      const int32_t* ref_idx, int32_t path_id, int32_t n_future, float* out, void* stream) {
    int rc = check_paths(h, "eb_tracking_error: null handle");
    if (rc) return rc;
    if (n < 0 || !xs || !ys || !phis || !vs || !out || n_future < 0) return fail(EB_EINVAL, "eb_tracking_error: bad argument");
    if (!ref_idx && (path_id < 0 || path_id >= h->pt.n_paths)) return fail(EB_EINVAL, "eb_tracking_error: bad path_id");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_tracking(h->cfg.task, n, h->pt, xs, ys, phis, vs, ref_idx, path_id, n_future, 10, out, nullptr,
                               nullptr, pick(h, stream)));
    return EB_OK;
}

int eb_veh_predict(eb_handle h, int32_t n_env, const float* veh, float* veh_out, void* stream) {
    if (!h || n_env < 0 || !veh || !veh_out) return fail(EB_EINVAL, "eb_veh_predict: bad argument");
    int rc = check_modes(h);
    if (rc) return rc;
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_veh_predict(n_env, h->cfg.n_veh, h->modes, veh, veh_out, pick(h, stream)));
    return EB_OK;
}

int eb_ss(eb_handle h, int32_t n_env, const float* obs, const float* actions, const int32_t* ref_idx,
          int32_t path_id, double lam, float* out, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_ss: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs || !actions || !out) return fail(EB_EINVAL, "eb_ss: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_ss(h->cfg.task, n_env, obs_dim(h->cfg), h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, obs,
                         actions, ref_idx, path_id, h->cfg.mode == EB_MODE_TRAINING, (float)(1.0 - lam), out,
                         pick(h, stream)));
    return EB_OK;
}

int eb_env_ego_step(eb_handle h, int32_t n, const float* ego, const float* actions, float* next_ego, float* params,
                    void* stream) {
    if (!h || n < 0 || !ego || !actions || !next_ego || !params) return fail(EB_EINVAL, "eb_env_ego_step: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_env_ego_step(n, ego, actions, next_ego, params, pick(h, stream)));
    return EB_OK;
}

// eb_debug_set_tile on the env-side kernels: 0 / 1 / 2 = 64- / 32- / 16-env tiles (every shape computes the same bits), else by batch size
static int forced_env_tile(const eb_handle_s* h) { return h->tile_variant == 0 ? 64 : h->tile_variant == 1 ? 32 : h->tile_variant == 2 ? 16 : 0; }

int eb_get_obs(eb_handle h, int32_t n_env, const float* ego, const int32_t* ref_idx, int32_t path_id,
               int32_t m_cand, const float* cand, const uint8_t* cand_mode, const uint8_t* v_light,
               const uint8_t* virtual_flag, const uint8_t* exit_id, const uint8_t* row_mask, float* obs_out, void* stream) {
    int rc = check_paths(h, "eb_get_obs: null handle");
    if (rc) return rc;
    rc = check_modes(h);
    if (rc) return rc;
    if (n_env < 0 || !ego || m_cand < 0 || m_cand > 256 || (m_cand > 0 && (!cand || !cand_mode)) || !obs_out)
        return fail(EB_EINVAL, "eb_get_obs: bad argument (m_cand <= 256)");
    if (!ref_idx && (path_id < 0 || path_id >= h->pt.n_paths)) return fail(EB_EINVAL, "eb_get_obs: bad path_id");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    // (exit ids live in device memory: the kernel masks them to 0..3 instead of a host-side range check)
    EB_HIP(eb::launch_get_obs(h->cfg.task, n_env, obs_dim(h->cfg), h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, ego,
                              ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, obs_out, pick(h, stream),
                              nullptr, nullptr, nullptr, exit_id, &h->xc, row_mask, nullptr, forced_env_tile(h), h->env_waves, h->trace,
                              h->trace_words, h->scan_one_trip));
    return EB_OK;
}

int eb_exit_frame(eb_handle h, int32_t n, const uint8_t* exit_id, int32_t inverse, const float* ego, float* ego_out,
                  void* stream) {
    if (!h || n < 0 || (n > 0 && (!exit_id || !ego || !ego_out))) return fail(EB_EINVAL, "eb_exit_frame: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_exit_frame(n, exit_id, inverse ? 1 : 0, h->xc, ego, ego_out, pick(h, stream)));
    return EB_OK;
}

int eb_judge_done(eb_handle h, int32_t n_env, const float* ego, const float* params, const float* obs,
                  int32_t m_cand, const float* cand, const uint8_t* cand_mode, const float* cand_lw,
                  const uint8_t* v_light, uint8_t* done_code, void* stream) {
    if (!h || n_env < 0 || !ego || !params || !obs || m_cand < 0 || (m_cand > 0 && (!cand || !cand_mode)) || !done_code)
        return fail(EB_EINVAL, "eb_judge_done: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_judge_done(h->cfg.task, n_env, obs_dim(h->cfg), ego, params, obs, m_cand, cand, cand_mode,
                                 cand_lw, v_light, done_code, pick(h, stream)));
    return EB_OK;
}

int eb_env_step(eb_handle h, eb_handle traffic, int32_t n_env, const float* obs, const float* actions,
                const int32_t* ref_idx, int32_t path_id, float* ego, float* params, int32_t m_cand, float* cand,
                const uint8_t* cand_mode, const float* cand_lw, const uint8_t* v_light, const uint8_t* virtual_flag,
                float* scaled_actions, float* out5, float* out_dict16, float* obs_out, uint8_t* done_code,
                const eb_respawn* respawn, const eb_auto_reset* auto_reset, const eb_flow_rule* flow,
                const eb_time_limit* time_limit, void* stream) {
    // every check first: an error return leaves ego / params / cand untouched
    if (!h || !traffic) return fail(EB_EINVAL, "eb_env_step: null handle");
    if (n_env < 0 || !obs || !actions || !ego || !params || !out5 || !obs_out || !done_code || obs == obs_out ||
        m_cand < 0 || m_cand > 256 || (m_cand > 0 && (!cand || !cand_mode)))
        return fail(EB_EINVAL, "eb_env_step: bad argument");
    if (respawn && (!respawn->entry || !(respawn->limit >= 0.0f) || m_cand < 1 || m_cand > 64))
        return fail(EB_EINVAL, "eb_env_step: bad respawn rule");
    if (traffic->cfg.n_veh != m_cand) return fail(EB_EINVAL, "eb_env_step: the traffic handle must have n_veh == m_cand");
    if (traffic->cfg.device != h->cfg.device) return fail(EB_EINVAL, "eb_env_step: the two handles live on different devices");
    int rc = check_paths(h, "eb_env_step: null handle");
    if (!rc) rc = check_modes(h);
    if (!rc) rc = check_modes(traffic);
    if (rc) return rc;
    if (!ref_idx && (path_id < 0 || path_id >= h->pt.n_paths)) return fail(EB_EINVAL, "eb_env_step: bad path_id");
    if (const eb_auto_reset* ar = auto_reset) {
        if (!flow && !ar->pool.entry)
            return fail(EB_EINVAL, "eb_env_step: auto_reset needs a traffic source to reset — the pool rule (auto_reset->pool.entry) or the flow rule of the call");
        if (m_cand < 1 || m_cand > 64) return fail(EB_EINVAL, "eb_env_step: auto_reset needs 1..64 candidates");
        if (!ref_idx || !virtual_flag || ar->ref_idx != ref_idx || ar->virtual_flag != virtual_flag || ar->v_light != v_light)
            return fail(EB_EINVAL, "eb_env_step: auto_reset rewrites the ref_idx / virtual_flag / v_light arrays of the call: they must be given and be the call's own");
        if (flow && (!ar->flow_cand_len || !ar->flow_phase0))
            return fail(EB_EINVAL, "eb_env_step: auto_reset over the flow source needs flow_cand_len and flow_phase0");
        if (ar->final_obs && (ar->final_obs == obs_out || ar->final_obs == obs))
            return fail(EB_EINVAL, "eb_env_step: final_obs must be an array of its own");
    }
    if (flow) {
        if (respawn) return fail(EB_EINVAL, "eb_env_step: the flow rule excludes respawn (the pool's rule)");
        if (flow->per_route < 1 || 12 * flow->per_route != m_cand || m_cand > 64 || !flow->active || !flow->timer || !flow->emitted ||
            !flow->sim_step || !flow->lane || !flow->period || !flow->v_max || !v_light || flow->v_light != v_light || flow->cand_mode != cand_mode)
            return fail(EB_EINVAL, "eb_env_step: bad flow rule (m_cand == 12 * per_route <= 64, every array given, cand_mode / v_light the call's own)");
    }
    if (time_limit && (!time_limit->episode_step || time_limit->max_episode_steps < 1))
        return fail(EB_EINVAL, "eb_env_step: bad time limit (episode_step given, max_episode_steps >= 1)");
    if (n_env == 0) return EB_OK;
    hipStream_t s = pick(h, stream);
    EB_HIP(hipSetDevice(h->cfg.device));
    const int D = obs_dim(h->cfg);
    if (eb::env_step_is_fused(D, h->cfg.n_veh, m_cand, cand, ego, actions, scaled_actions, params, flow != nullptr)) {
        // the whole step — the six calls and the pool's re-entry — as ONE launch (csrc/eb_env_step.hip)
        eb::EnvStepArgs A;
        std::memset(&A, 0, sizeof A);
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
        A.n_env = n_env; A.D = D; A.n_future = h->cfg.n_future; A.NV = h->cfg.n_veh; A.m_cand = m_cand; A.path_id = path_id;
        A.d_magic = magic(D); A.m_magic = magic(m_cand); A.nv_magic = magic(h->cfg.n_veh);
        A.pt = h->pt; A.modes = h->modes;
        std::memcpy(A.tturn.t, traffic->modes.turn, sizeof A.tturn.t);
        eb::env_step_slot_plan(h->modes, h->cfg.n_veh, A);
        A.obs = obs; A.raw = actions; A.ref_idx = ref_idx; A.ego = ego; A.params = params; A.cand = cand; A.cand_mode = cand_mode;
        A.cand_lw = cand_lw; A.v_light = v_light; A.virtual_flag = virtual_flag; A.scaled = scaled_actions; A.out5 = out5;
        A.d16 = out_dict16; A.obs_out = obs_out; A.done_code = done_code;
        A.trace = h->trace; A.trace_words = h->trace_words;
        A.tile_envs = forced_env_tile(h); A.waves = h->env_waves; A.scan_one_trip = h->scan_one_trip;
        A.by_progress = h->sched_progress;   // (-1: launch_env_step decides by the grid)
        if (respawn) {
            A.respawn_entry = respawn->entry; A.limit = respawn->limit; A.span = respawn->span; A.v_max = respawn->v_max;
            A.seed = respawn->seed; A.counter = respawn->counter;
        }
        if (const eb_auto_reset* ar = auto_reset) {   // the rows this step finishes are reset by their own block, same launch
            A.auto_reset = 1; A.training = ar->training ? 1 : 0; A.reset_seed = ar->seed; A.reset_counter = ar->counter;
            A.ref_idx_out = ar->ref_idx; A.virtual_out = ar->virtual_flag; A.v_light_out = ar->v_light; A.final_obs = ar->final_obs;
            A.pool_entry = ar->pool.entry; A.pool_span = ar->pool.span; A.pool_v_max = ar->pool.v_max; A.edge_span = ar->pool.edge_span;
            A.pool_seed = ar->pool.seed; A.pool_counter = ar->pool.counter;
            A.flow_cand_len = ar->flow_cand_len; A.flow_phase0 = ar->flow_phase0; A.flow_random_phase = ar->flow_random_phase ? 1 : 0;
            A.flow_reset_seed = ar->flow_seed; A.flow_reset_counter = ar->flow_counter;
        }
        if (flow) {   // the flow source's step rides on the way out of the same launch
            A.flow_on = 1; A.flow_K = flow->per_route; A.flow_active = flow->active; A.flow_timer = flow->timer; A.flow_emitted = flow->emitted;
            A.flow_sim_step = flow->sim_step; A.flow_lane = flow->lane; A.flow_period = flow->period; A.flow_v_max = flow->v_max;
            A.flow_dt = flow->dt; A.flow_exit_range = flow->exit_range; A.flow_accel = flow->accel; A.flow_lane_len = flow->lane_len;
            A.flow_light_cycle = flow->light_cycle; A.seed = flow->seed; A.counter = flow->counter;
            A.flow_mode_out = flow->cand_mode; A.v_light_out = flow->v_light;
            A.k_magic = magic(flow->per_route);
        }
        if (time_limit) { A.episode_step = time_limit->episode_step; A.max_episode_steps = time_limit->max_episode_steps; }
        EB_HIP(eb::launch_env_step(h->cfg.task, A, s));
        return EB_OK;
    }
    // separate launches (no candidates, a tile that does not fit the LDS, an unaligned candidate buffer)
    // (scratch per call, allocated and released in stream order: two streams may drive one handle through this path)
    float* scaled = scaled_actions;
    float* own_scaled = nullptr;
    if (!scaled) {
        EB_HIP(hipMallocAsync(reinterpret_cast<void**>(&own_scaled), (size_t)n_env * 2 * sizeof(float), s));
        scaled = own_scaled;
    }
    struct Release {   // on every return path below, after the launches that read it
        float* p; hipStream_t s;
        ~Release() { if (p) (void)hipFreeAsync(p, s); }
    } release{own_scaled, s};
    // E2E:133-135 in one launch: action scaling, reward on the current obs, ego step in place (the same device
    // functions eb_action_transform / eb_compute_rewards / eb_env_ego_step run)
    EB_HIP(eb::launch_env_pre(h->cfg.task, n_env, D, h->cfg.n_future, h->cfg.n_veh, obs, actions,
                              scaled, out5, out_dict16, ego, params, s));
    if (m_cand > 0) EB_HIP(eb::launch_veh_predict(n_env, m_cand, traffic->modes, cand, cand, s));   /* TRF:220-238's role */
    if (m_cand > 0 && eb::get_obs_is_staged(D, m_cand, cand)) {
        // E2E:140-141 in one launch: the observation kernel keeps the tile's candidates and the new delta_y in LDS and
        // appends _judge_done (the same device functions eb_judge_done runs)
        EB_HIP(eb::launch_get_obs(h->cfg.task, n_env, D, h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, ego,
                                  ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, obs_out, s, params,
                                  cand_lw, done_code));
    } else {
        EB_HIP(eb::launch_get_obs(h->cfg.task, n_env, D, h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, ego,
                                  ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, obs_out, s));   /* E2E:140 */
        EB_HIP(eb::launch_judge_done(h->cfg.task, n_env, D, ego, params, obs_out, m_cand, cand, cand_mode,
                                     cand_lw, v_light, done_code, s));                                          /* E2E:141 */
    }
    if (time_limit) EB_HIP(eb::launch_time_limit(n_env, time_limit->episode_step, time_limit->max_episode_steps, done_code, s));
    if (respawn)
        EB_HIP(eb::launch_traffic_respawn(n_env, m_cand, cand, respawn->entry, respawn->limit, respawn->span, respawn->v_max,
                                          respawn->seed, respawn->counter, nullptr, nullptr, s));
    if (flow && auto_reset) {   // the flow source: its step first, then the masked reset's launches (eb_env_reset, the source's own reset, obs, flags)
        const eb_auto_reset* ar = auto_reset;
        EB_HIP(eb::launch_traffic_flow_step(n_env, flow->per_route, cand, flow->active, flow->timer, flow->emitted, flow->sim_step, flow->lane,
                                            flow->period, flow->v_max, flow->dt, flow->exit_range, flow->accel, flow->lane_len,
                                            flow->light_cycle, flow->seed, flow->counter, flow->cand_mode, flow->v_light, s));
        if (ar->final_obs) EB_HIP(eb::launch_copy_rows_masked(n_env, D, done_code, obs_out, ar->final_obs, s));
        uint8_t* d_vnext = nullptr;
        EB_HIP(hipMallocAsync(reinterpret_cast<void**>(&d_vnext), (size_t)n_env, s));
        struct ReleaseV { uint8_t* p; hipStream_t s; ~ReleaseV() { (void)hipFreeAsync(p, s); } } release_v{d_vnext, s};
        EB_HIP(eb::launch_env_reset(h->cfg.task, n_env, h->pt, done_code, ar->seed, ar->counter, ar->training ? 1 : 0, ego, params, ar->ref_idx,
                                    d_vnext, nullptr, s));
        EB_HIP(eb::launch_traffic_flow_reset(n_env, flow->per_route, done_code, ego, cand, flow->active, flow->timer, flow->emitted, flow->sim_step,
                                             ar->flow_phase0, flow->lane, flow->period, flow->v_max, ar->flow_cand_len, flow->lane_len,
                                             ar->flow_random_phase ? 1 : 0, ar->training ? 1 : 0, ar->flow_seed, ar->flow_counter,
                                             flow->cand_mode, flow->v_light, s));
        EB_HIP(eb::launch_get_obs(h->cfg.task, n_env, D, h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, ego, ar->ref_idx, 0, m_cand, cand,
                                  flow->cand_mode, flow->v_light, ar->virtual_flag, obs_out, s, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  done_code));
        EB_HIP(eb::launch_flag_swap(n_env, done_code, d_vnext, ar->virtual_flag, s));
        return EB_OK;
    }
    if (const eb_auto_reset* ar = auto_reset) {   // the same composition the header spells out, as launches of their own
        if (ar->final_obs) EB_HIP(eb::launch_copy_rows_masked(n_env, D, done_code, obs_out, ar->final_obs, s));
        return eb_env_reset_pool(h, traffic, n_env, done_code, ar->seed, ar->counter, ar->training, ego, params, ar->ref_idx,
                                 ar->virtual_flag, ar->v_light, nullptr, nullptr, m_cand, cand, cand_mode, &ar->pool, obs_out, nullptr,
                                 nullptr, stream);   // (the time limit above has restarted the finished envs' counts)
    }
    if (flow)
        EB_HIP(eb::launch_traffic_flow_step(n_env, flow->per_route, cand, flow->active, flow->timer, flow->emitted, flow->sim_step, flow->lane,
                                            flow->period, flow->v_max, flow->dt, flow->exit_range, flow->accel, flow->lane_len,
                                            flow->light_cycle, flow->seed, flow->counter, flow->cand_mode, flow->v_light, s));
    return EB_OK;
}

int eb_env_reset(eb_handle h, int32_t n_env, const uint8_t* mask, uint64_t seed, uint64_t counter, int32_t training,
                 float* ego, float* params, int32_t* ref_idx, uint8_t* virtual_next, uint8_t* done_code, int32_t* episode_step,
                 void* stream) {
    int rc = check_paths(h, "eb_env_reset: null handle");
    if (rc) return rc;
    if (n_env < 0 || (n_env > 0 && (!ego || !params || !ref_idx))) return fail(EB_EINVAL, "eb_env_reset: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_env_reset(h->cfg.task, n_env, h->pt, mask, seed, counter, training ? 1 : 0, ego, params, ref_idx,
                                virtual_next, done_code, pick(h, stream), nullptr, episode_step));
    return EB_OK;
}

int eb_ego_dynamics(eb_handle h, int32_t n, const float* ego, const float* params, float* out, void* stream) {
    if (!h || n < 0 || (n > 0 && (!ego || !params || !out))) return fail(EB_EINVAL, "eb_ego_dynamics: bad argument");
    if (n == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_ego_dynamics(n, ego, params, out, pick(h, stream)));
    return EB_OK;
}

int eb_env_reset_pool(eb_handle h, eb_handle traffic, int32_t n_env, const uint8_t* mask, uint64_t seed, uint64_t counter,
                      int32_t training, float* ego, float* params, int32_t* ref_idx, uint8_t* virtual_flag, uint8_t* v_light,
                      uint8_t* done_code, int32_t* episode_step, int32_t m_cand, float* cand, const uint8_t* cand_mode,
                      const eb_respawn* pool, float* obs, const float* obs_src, const uint8_t* done_src, void* stream) {
    if (!h || !traffic || !pool || !pool->entry) return fail(EB_EINVAL, "eb_env_reset_pool: null argument");
    if (n_env < 0 || m_cand < 1 || m_cand > 64 || (n_env > 0 && (!ego || !params || !ref_idx || !virtual_flag || !cand || !cand_mode || !obs)))
        return fail(EB_EINVAL, "eb_env_reset_pool: bad argument");
    if (mask && (mask == done_code || mask == virtual_flag || mask == v_light))
        return fail(EB_EINVAL, "eb_env_reset_pool: mask must not be one of the arrays the call writes (pass the done codes as mask and done_src, a fresh array as done_code)");
    if (traffic->cfg.n_veh != m_cand) return fail(EB_EINVAL, "eb_env_reset_pool: the traffic handle must have n_veh == m_cand");
    if (traffic->cfg.device != h->cfg.device) return fail(EB_EINVAL, "eb_env_reset_pool: the two handles live on different devices");
    int rc = check_paths(h, "eb_env_reset_pool: null handle");
    if (!rc) rc = check_modes(h);
    if (rc) return rc;
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    if (eb::env_step_is_fused(obs_dim(h->cfg), h->cfg.n_veh, m_cand, cand, ego, nullptr, nullptr, params)) {
        // ONE launch (csrc/eb_env_step.hip, env_reset_pool_kernel): a tile's masked rows are drawn, their pool re-entered clear of
        // the new ego, their observation built from the state still in LDS, their flag swapped — the same arithmetic, in the
        // same order per row, as the four launches below
        const eb::EnvResetArgs R{seed, counter, training ? 1 : 0, params, ref_idx, virtual_flag, v_light, done_code, pool->entry,
                                 pool->span, pool->v_max, pool->edge_span, pool->seed, pool->counter,
                                 mask && obs_src != obs ? obs_src : nullptr, mask && done_code && done_src != done_code ? done_src : nullptr,
                                 episode_step};
        EB_HIP(eb::launch_get_obs(h->cfg.task, n_env, obs_dim(h->cfg), h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, ego, ref_idx, 0,
                                  m_cand, cand, cand_mode, nullptr, virtual_flag, obs, pick(h, stream), nullptr, nullptr, nullptr, nullptr,
                                  nullptr, mask, &R, forced_env_tile(h), h->env_waves, h->trace, h->trace_words, h->scan_one_trip));
        return EB_OK;
    }
    hipStream_t s = pick(h, stream);
    uint8_t* d_vnext = nullptr;   // the flags eb_env_reset draws, until they are swapped in: per call, in stream order
    EB_HIP(hipMallocAsync(reinterpret_cast<void**>(&d_vnext), (size_t)n_env, s));
    struct Release {
        uint8_t* p; hipStream_t s;
        ~Release() { (void)hipFreeAsync(p, s); }
    } release{d_vnext, s};
    if (mask && obs_src && obs_src != obs)       // the rows outside the mask: carried over from the caller's previous arrays
        EB_HIP(hipMemcpyAsync(obs, obs_src, (size_t)n_env * obs_dim(h->cfg) * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (mask && done_src && done_code && done_src != done_code)
        EB_HIP(hipMemcpyAsync(done_code, done_src, (size_t)n_env, hipMemcpyDeviceToDevice, s));
    // (an unaligned candidate buffer or a tile that does not fit the LDS) four launches behind one call: state + flags, pool
    // re-entry (clear of the ego), masked observation, flag swap
    EB_HIP(eb::launch_env_reset(h->cfg.task, n_env, h->pt, mask, seed, counter, training ? 1 : 0, ego, params, ref_idx, d_vnext,
                                done_code, s, v_light, episode_step));
    EB_HIP(eb::launch_traffic_respawn(n_env, m_cand, cand, pool->entry, -1.0f, pool->span, pool->v_max, pool->seed, pool->counter, mask,
                                      nullptr, s, ego, pool->edge_span));
    EB_HIP(eb::launch_get_obs(h->cfg.task, n_env, obs_dim(h->cfg), h->cfg.n_future, h->cfg.n_veh, h->pt, h->modes, ego, ref_idx, 0,
                              m_cand, cand, cand_mode, v_light, virtual_flag, obs, s, nullptr, nullptr, nullptr, nullptr, nullptr, mask));
    EB_HIP(eb::launch_flag_swap(n_env, mask, d_vnext, virtual_flag, s));
    return EB_OK;
}

// Diagnostics (include/envbuild.h, last section): device buffer of [n_waves][8] int64 that the rollout kernel
// fills with per-wave wall-clock marks (100 MHz) — scripts/trace_rollout.py.  NULL switches it off.
int eb_debug_set_trace(eb_handle h, long long* device_buf, int64_t capacity_words) {
    if (!h || (device_buf && capacity_words < 0)) return fail(EB_EINVAL, "eb_debug_set_trace: bad argument");
    h->trace = device_buf;
    h->trace_words = device_buf ? capacity_words : 0;
    return EB_OK;
}

int eb_debug_set_stage_paths(eb_handle h, int32_t mode) {
    if (!h || mode < -1 || mode > 1) return fail(EB_EINVAL, "eb_debug_set_stage_paths: bad argument (-1 = by grid size, 0 = off, 1 = on)");
    h->stage_paths = mode;
    return EB_OK;
}

// Self-check of the closest-point levels (host; the tables and cell words the kernels read): positions sampled in every cell of every
// level — uniformly, and pressed against the cell's edges and corners — are mapped to their cell with the kernels' fp32 expression,
// the reference's first minimum (DAM:712-714: fp32 squares, strict '<', index order) is taken over the WHOLE stride-10 table, and the
// cell's range(s) must hold it.  -> *n_checked positions, *n_bad of them outside their cell's ranges.
int eb_debug_check_grids(eb_handle h, int32_t samples_per_cell, uint64_t seed, int64_t* n_checked, int64_t* n_bad) {
    if (!h || !h->grids_ref || samples_per_cell < 1 || !n_checked || !n_bad) return fail(EB_EINVAL, "eb_debug_check_grids: bad argument (paths set?)");
    const GridSet& g = **static_cast<std::shared_ptr<const GridSet>*>(h->grids_ref);
    const int n_paths = (int)g.key_len.size();
    int64_t checked = 0, bad = 0;
    uint64_t rs = seed * 0x9E3779B97F4A7C15ull + 1;
    auto u01 = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) * (1.0 / 9007199254740992.0); };
    for (int l = 0; l < 4; ++l) {
        const CellGrid& cg = l == 0 ? g.fine : g.lvl[l - 1];
        const double cell = l == 0 ? 1.0 / (double)eb::CELL_INV : GRID_LVL_CELL[l - 1];
        const float inv = (float)(1.0 / cell), gx0 = (float)cg.x0, gy0 = (float)cg.y0;
        const uint32_t* words = g.cells.data() + (l == 0 ? 0 : g.lvl_off[l - 1]);
        size_t off = 0;
        for (int k = 0; k < n_paths; ++k) {
            const float* r = g.key.data() + 2 * off;
            const int n = g.key_len[k];
            off += (size_t)n;
            for (int iy = 0; iy < cg.ny; ++iy)
                for (int ix = 0; ix < cg.nx; ++ix)
                    for (int sidx = 0; sidx < samples_per_cell; ++sidx) {
                        double ux = u01(), uy = u01();
                        if (sidx % 3 == 1) { ux = ux < 0.5 ? ux * 1e-4 : 1.0 - ux * 1e-4; }                    // against an edge
                        if (sidx % 3 == 2) { ux = ux < 0.5 ? ux * 1e-5 : 1.0 - ux * 1e-5; uy = uy < 0.5 ? uy * 1e-5 : 1.0 - uy * 1e-5; }   // into a corner
                        const float px = (float)(cg.x0 + (ix + ux) * cell), py = (float)(cg.y0 + (iy + uy) * cell);
                        const float fx = (px - gx0) * inv, fy = (py - gy0) * inv;                                // the kernels' mapping
                        if (!(fx >= 0.0f && fx < (float)cg.nx && fy >= 0.0f && fy < (float)cg.ny)) continue;
                        const uint32_t c = words[((size_t)k * cg.ny + (int)fy) * cg.nx + (int)fx];
                        float best = INFINITY;
                        int bi = 0;
                        for (int i = 0; i < n; ++i) {
                            const float dx = px - r[2 * i], dy = py - r[2 * i + 1];
                            const float d = dx * dx + dy * dy;
                            if (d < best) { best = d; bi = i; }
                        }
                        bool ok;
                        if (l == 0) ok = c == 0xffffffffu || (bi >= (int)(c & 0xffffu) && bi <= (int)(c >> 16));
                        else if (c >> 31) ok = c == 0xffffffffu || (bi >= (int)(c & 0x1ffu) && bi <= (int)((c >> 9) & 0x1ffu));
                        else {
                            const int lo = (int)(c & 0x1ffu), hi = lo + (int)((c >> 9) & 0x3fu);
                            const int lo2 = (int)((c >> 15) & 0x1ffu), hi2 = lo2 + (int)((c >> 24) & 0x3fu);
                            ok = (bi >= lo && bi <= hi) || (((c >> 30) & 1u) && bi >= lo2 && bi <= hi2);
                        }
                        ++checked;
                        bad += ok ? 0 : 1;
                    }
        }
    }
    *n_checked = checked;
    *n_bad = bad;
    return EB_OK;
}

int eb_debug_set_scan_prefetch(eb_handle h, int32_t on) {
    if (!h) return fail(EB_EINVAL, "eb_debug_set_scan_prefetch: null handle");
    h->scan_one_trip = on ? 0 : 1;
    return EB_OK;
}

int eb_debug_set_rollout_sched(eb_handle h, int32_t rolling, int32_t by_progress) {
    if (!h || rolling < -1 || rolling > 1 || by_progress < -1 || by_progress > 1)
        return fail(EB_EINVAL, "eb_debug_set_rollout_sched: bad argument (-1 = by grid size, 0, 1)");
    h->sched_rolling = rolling; h->sched_progress = by_progress;
    return EB_OK;
}

int eb_debug_rollout_plan(eb_handle h, int32_t n_env, int32_t* out4) {
    if (!h || n_env < 1 || !out4) return fail(EB_EINVAL, "eb_debug_rollout_plan: bad argument");
    const int variant = pick_variant(h, n_env), e = envs_per_tile(h, variant), grid = (n_env + e - 1) / e;
    int rolling = 0, by_progress = 0;
    rollout_sched(h, variant, grid, e, &rolling, &by_progress);
    out4[0] = variant; out4[1] = grid; out4[2] = rolling; out4[3] = by_progress;
    return EB_OK;
}

int eb_debug_set_env_waves(eb_handle h, int32_t waves) {
    if (!h || (waves != 0 && waves != 4 && waves != 8)) return fail(EB_EINVAL, "eb_debug_set_env_waves: bad argument (0 = by grid size, 4, 8)");
    h->env_waves = waves;
    return EB_OK;
}

// Test / tuning aid: force the rollout kernel's tile shape — 0: 2048-record
// tiles (4 record waves x 8 records per lane), 1: 1024 (4 x 4), 2: 256 (1 x 4); -1: pick by batch size.
// Every shape computes the same bits; the tests run all of them at small sizes.
int eb_debug_set_tile(eb_handle h, int32_t variant) {
    if (!h || variant < -1 || variant > 2) return fail(EB_EINVAL, "eb_debug_set_tile: bad argument");
    h->tile_variant = variant;
    return EB_OK;
}

// Test aid: 1 = eb_rollout_tape[_f16] as H per-step launches, 0 = one tape-kernel launch.
int eb_debug_set_tape_stepwise(eb_handle h, int32_t on) {
    if (!h) return fail(EB_EINVAL, "eb_debug_set_tape_stepwise: null handle");
    h->tape_stepwise = on ? 1 : 0;
    return EB_OK;
}

int eb_episode_summary(eb_handle h, int32_t n_env, int32_t horizon, const float* out5_steps, const float* obs_final,
                       float* out8, void* stream) {
    if (!h || n_env < 0 || horizon < 0 || !out8 || (n_env > 0 && horizon > 0 && !out5_steps) || (n_env > 0 && !obs_final))
        return fail(EB_EINVAL, "eb_episode_summary: bad argument");
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_summary(n_env, horizon, obs_dim(h->cfg), out5_steps, obs_final, h->d_partials,
                              eb::SUMMARY_MAX_PARTS, out8, pick(h, stream)));
    return EB_OK;
}

}  // extern "C"

struct eb_plan_s {
    eb_handle h;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

struct eb_event_s {
    hipEvent_t ev;
    int device;
};

extern "C" {

int eb_plan_create(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                   const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out, float* out5_steps,
                   float* summary8, void* acc, eb_plan* out) {
    if (!out) return fail(EB_EINVAL, "eb_plan_create: null argument");
    *out = nullptr;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_plan_create: null handle");
    if (rc) return rc;
    if (n_env < 1 || horizon < 1 || !obs_in || !action_tape || !obs_work || !obs_out || !out5_steps)
        return fail(EB_EINVAL, "eb_plan_create: bad argument (n_env >= 1, horizon >= 1, non-null buffers)");
    if (obs_work == obs_out || obs_in == obs_work || obs_in == obs_out)
        return fail(EB_EINVAL, "eb_plan_create: obs_in, obs_work and obs_out must be distinct buffers");
    if (acc && ((uintptr_t)acc & 15) != 0) return fail(EB_EINVAL, "eb_plan_create: acc must be 16-byte aligned");
    EB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { return fail_hip("hipStreamCreateWithFlags", e); }
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { (void)hipStreamDestroy(cs); return fail_hip("hipStreamBeginCapture", e); }
    // the plan is the CLOSED-LOOP form: one per-step launch per rollout_out, H of them in a graph.  With a caller's accumulator they
    // are the accumulating launches and the summary (if asked for) is the fold behind them; without one the summary is
    // eb_episode_summary's second pass over out5 — the faster of the two on this GPU (profiles/r5_ab_acc_summary.txt)
    rc = rollout_tape_stepwise(h, n_env, horizon, obs_in, action_tape, ref_idx, path_id, obs_work, obs_out, out5_steps, cs, 0, acc);
    if (rc == EB_OK && summary8)
        rc = acc ? eb_episode_acc_finish(h, n_env, horizon, acc, summary8, cs)
                 : eb_episode_summary(h, n_env, horizon, out5_steps, obs_out, summary8, cs);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(cs, &graph);
    (void)hipStreamDestroy(cs);
    if (rc != EB_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { return fail_hip("hipStreamEndCapture", e); }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); return fail_hip("hipGraphInstantiate", e); }
    eb_plan p = new (std::nothrow) eb_plan_s();
    if (!p) {
        (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
       
        return fail(EB_ENOMEM, "eb_plan_create: out of memory");
    }
    p->h = h; p->graph = graph; p->exec = exec;
    *out = p;
    return EB_OK;
}

int eb_plan_launch(eb_plan p, void* stream) {
    if (!p) return fail(EB_EINVAL, "eb_plan_launch: null plan");
    EB_HIP(hipSetDevice(p->h->cfg.device));
    EB_HIP(hipGraphLaunch(p->exec, (hipStream_t)stream));
    return EB_OK;
}

int eb_plan_destroy(eb_plan p) {
    if (!p) return EB_OK;
    (void)hipSetDevice(p->h->cfg.device);
    (void)hipDeviceSynchronize();
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    delete p;
    return EB_OK;
}

int eb_event_create(eb_handle h, eb_event* out) {
    if (!h || !out) return fail(EB_EINVAL, "eb_event_create: null argument");
    EB_HIP(hipSetDevice(h->cfg.device));
    hipEvent_t ev;
    EB_HIP(hipEventCreate(&ev));
    eb_event e = new (std::nothrow) eb_event_s();
    if (!e) { (void)hipEventDestroy(ev); return fail(EB_ENOMEM, "eb_event_create: out of memory"); }
    e->ev = ev; e->device = h->cfg.device;
    *out = e;
    return EB_OK;
}

int eb_event_record(eb_event e, void* stream) {
    if (!e) return fail(EB_EINVAL, "eb_event_record: null event");
    EB_HIP(hipSetDevice(e->device));
    EB_HIP(hipEventRecord(e->ev, (hipStream_t)stream));
    return EB_OK;
}

int eb_event_elapsed_ms(eb_event start, eb_event stop, float* ms) {
    if (!start || !stop || !ms) return fail(EB_EINVAL, "eb_event_elapsed_ms: null argument");
    EB_HIP(hipSetDevice(stop->device));
    EB_HIP(hipEventSynchronize(stop->ev));
    EB_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return EB_OK;
}

int eb_event_destroy(eb_event e) {
    if (!e) return EB_OK;
    (void)hipSetDevice(e->device);
    (void)hipEventDestroy(e->ev);
    delete e;
    return EB_OK;
}

}  // extern "C"

// ---- policy network + shield ----
struct eb_mlp_s {
    eb_mlp_config cfg;
    int units;                       // padded hidden width
    int k_pad0;                      // padded obs_dim
    float* d_w[EB_MLP_MAX_HIDDEN + 1];
    float* d_b[EB_MLP_MAX_HIDDEN + 1];
    float* d_scale;
    bool has_scale;
    unsigned layers_set;
};

static int mlp_layer_dims(const eb_mlp_s* m, int layer, int* k_real, int* cols_real, int* k_pad, int* col_tiles) {
    const bool out = layer == m->cfg.n_hidden;
    *k_real = layer == 0 ? m->cfg.obs_dim : m->cfg.n_units;
    *cols_real = out ? m->cfg.out_dim : m->cfg.n_units;
    *k_pad = layer == 0 ? m->k_pad0 : m->units;
    *col_tiles = out ? 1 : m->units / 32;
    return 0;
}

static int mlp_args(eb_mlp m, int32_t n, const float* obs, float* out, int head, float action_range, eb::MlpArgs* A,
                    const char* who) {
    if (!m) return fail(EB_EINVAL, who);
    if (n < 0 || (n > 0 && (!obs || !out))) return fail(EB_EINVAL, "eb_mlp: bad argument");
    if (m->layers_set != (1u << (m->cfg.n_hidden + 1)) - 1u) return fail(EB_ESTATE, "eb_mlp: eb_mlp_set_layer has not been called for every layer");
    if (head == eb::MLP_HEAD_ACTION && (m->cfg.out_dim < 2 || (m->cfg.out_dim & 1)))
        return fail(EB_EINVAL, "eb_policy_run_batch: out_dim must be 2 * act_dim");
    A->obs = obs; A->scale = m->has_scale ? m->d_scale : nullptr; A->out = out;
    A->n = n; A->obs_dim = m->cfg.obs_dim; A->n_hidden = m->cfg.n_hidden; A->units = m->units;
    A->out_dim = m->cfg.out_dim; A->hidden_act = m->cfg.hidden_act; A->out_act = m->cfg.out_act; A->head = head;
    A->action_range = action_range;
    A->row_stride = std::max(m->k_pad0, m->units) + 4;
    for (int L = 0; L < m->cfg.n_hidden; ++L) {
        A->hid[L].w = m->d_w[L]; A->hid[L].b = m->d_b[L]; A->hid[L].k_pad = L == 0 ? m->k_pad0 : m->units; A->hid[L].pad_ = 0;
    }
    for (int L = m->cfg.n_hidden; L < eb::MLP_MAX_HIDDEN; ++L) A->hid[L] = eb::MlpLayer{nullptr, nullptr, 0, 0};
    A->outl.w = m->d_w[m->cfg.n_hidden]; A->outl.b = m->d_b[m->cfg.n_hidden]; A->outl.k_pad = m->units; A->outl.pad_ = 0;
    return EB_OK;
}

extern "C" {

int eb_traffic_respawn(eb_handle h, int32_t n_env, int32_t m_cand, float* cand, const float* entry, float limit,
                       float span, float v_max, uint64_t seed, uint64_t counter, const uint8_t* env_mask,
                       uint8_t* respawned, const float* ego, float edge_span, void* stream) {
    if (!h || n_env < 0 || m_cand < 1 || m_cand > 64 || (n_env > 0 && (!cand || !entry)))
        return fail(EB_EINVAL, "eb_traffic_respawn: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_traffic_respawn(n_env, m_cand, cand, entry, limit, span, v_max, seed, counter, env_mask, respawned,
                                      pick(h, stream), ego, edge_span));
    return EB_OK;
}

int eb_traffic_flow_reset(eb_handle h, int32_t n_env, int32_t per_route, const uint8_t* mask, const float* ego,
                          float* cand, uint8_t* active, float* timer, int32_t* emitted, int32_t* sim_step,
                          uint8_t* phase0, const float* lane, const float* period, const float* v_max,
                          const float* cand_len, float lane_len, int32_t random_phase, int32_t training,
                          uint64_t seed, uint64_t counter, uint8_t* cand_mode, uint8_t* v_light, void* stream) {
    if (!h || n_env < 0 || per_route < 1 || per_route * 12 > 64 ||
        (n_env > 0 && (!ego || !cand || !active || !timer || !emitted || !sim_step || !phase0 || !lane || !period || !v_max ||
                       !cand_len || !cand_mode || !v_light)))
        return fail(EB_EINVAL, "eb_traffic_flow_reset: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_traffic_flow_reset(n_env, per_route, mask, ego, cand, active, timer, emitted, sim_step, phase0, lane,
                                         period, v_max, cand_len, lane_len, random_phase ? 1 : 0, training ? 1 : 0, seed,
                                         counter, cand_mode, v_light, pick(h, stream)));
    return EB_OK;
}

int eb_traffic_flow_step(eb_handle h, int32_t n_env, int32_t per_route, float* cand, uint8_t* active, float* timer,
                         int32_t* emitted, int32_t* sim_step, const float* lane, const float* period,
                         const float* v_max, float dt, float exit_range, float accel, float lane_len,
                         int32_t light_cycle, uint64_t seed, uint64_t counter, uint8_t* cand_mode, uint8_t* v_light,
                         void* stream) {
    if (!h || n_env < 0 || per_route < 1 || per_route * 12 > 64 || !(dt > 0.0f) ||
        (n_env > 0 && (!cand || !active || !timer || !emitted || !sim_step || !lane || !period || !v_max || !cand_mode || !v_light)))
        return fail(EB_EINVAL, "eb_traffic_flow_step: bad argument");
    if (n_env == 0) return EB_OK;
    EB_HIP(hipSetDevice(h->cfg.device));
    EB_HIP(eb::launch_traffic_flow_step(n_env, per_route, cand, active, timer, emitted, sim_step, lane, period, v_max, dt,
                                        exit_range, accel, lane_len, light_cycle, seed, counter, cand_mode, v_light,
                                        pick(h, stream)));
    return EB_OK;
}

int eb_mlp_create(const eb_mlp_config* cfg, eb_mlp* out) {
    if (!cfg || !out) return fail(EB_EINVAL, "eb_mlp_create: null argument");
    if (cfg->abi_version != EB_ABI_VERSION) return fail(EB_EINVAL, "eb_mlp_create: ABI version mismatch");
    if (cfg->obs_dim < 1 || cfg->n_hidden < 1 || cfg->n_hidden > EB_MLP_MAX_HIDDEN || cfg->n_units < 1 ||
        cfg->n_units > EB_MLP_MAX_UNITS || cfg->out_dim < 1 || cfg->out_dim > 32)
        return fail(EB_EINVAL, "eb_mlp_create: bad dimensions");
    if (cfg->hidden_act < EB_ACT_LINEAR || cfg->hidden_act > EB_ACT_TANH || cfg->out_act < EB_ACT_LINEAR || cfg->out_act > EB_ACT_TANH)
        return fail(EB_EINVAL, "eb_mlp_create: unknown activation");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(EB_EDEVICE, "eb_mlp_create: no HIP device (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(EB_EINVAL, "eb_mlp_create: bad device ordinal");
    EB_HIP(hipSetDevice(cfg->device));
    eb_mlp_s* m = new (std::nothrow) eb_mlp_s();
    if (!m) return fail(EB_ENOMEM, "eb_mlp_create: out of memory");
    m->cfg = *cfg;
    m->units = eb::mlp_padded_units(cfg->n_units);
    m->k_pad0 = (cfg->obs_dim + 7) / 8 * 8;
    if ((size_t)eb::MLP_ROWS * (std::max(m->k_pad0, m->units) + 4) * sizeof(float) > 160 * 1024) {
        delete m;
        return fail(EB_EINVAL, "eb_mlp_create: obs_dim too large for the LDS activation buffer");
    }
    for (int L = 0; L <= cfg->n_hidden; ++L) {
        int kr, cr, kp, ct;
        mlp_layer_dims(m, L, &kr, &cr, &kp, &ct);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_w[L]), sizeof(float) * (size_t)kp * ct * 32);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&m->d_b[L]), sizeof(float) * ct * 32);
        if (e != hipSuccess) { eb_mlp_destroy(m); return fail_hip("hipMalloc(mlp layer)", e); }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_scale), sizeof(float) * cfg->obs_dim);
    if (e != hipSuccess) { eb_mlp_destroy(m); return fail_hip("hipMalloc(obs scale)", e); }
    *out = m;
    return EB_OK;
}

int eb_mlp_destroy(eb_mlp m) {
    if (!m) return EB_OK;
    for (int L = 0; L <= EB_MLP_MAX_HIDDEN; ++L) {
        if (m->d_w[L]) (void)hipFree(m->d_w[L]);
        if (m->d_b[L]) (void)hipFree(m->d_b[L]);
    }
    if (m->d_scale) (void)hipFree(m->d_scale);
    delete m;
    return EB_OK;
}

int eb_mlp_set_layer(eb_mlp m, int32_t layer, const float* kernel, const float* bias) {
    if (!m || !kernel || !bias || layer < 0 || layer > m->cfg.n_hidden) return fail(EB_EINVAL, "eb_mlp_set_layer: bad argument");
    int kr, cr, kp, ct;
    mlp_layer_dims(m, layer, &kr, &cr, &kp, &ct);
    std::vector<float> wp((size_t)kp * ct * 32), bp((size_t)ct * 32, 0.0f);
    if (layer == m->cfg.n_hidden) eb::pack_weights16(kernel, kr, cr, kp, wp.data());   // (<= 2 tiles of 16: fits the 32-column buffer)
    else eb::pack_weights(kernel, kr, cr, kp, ct, wp.data());
    std::copy(bias, bias + cr, bp.begin());
    EB_HIP(hipSetDevice(m->cfg.device));
    EB_HIP(hipMemcpy(m->d_w[layer], wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
    EB_HIP(hipMemcpy(m->d_b[layer], bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
    m->layers_set |= 1u << layer;
    return EB_OK;
}

int eb_mlp_set_obs_scale(eb_mlp m, const float* scale) {
    if (!m) return fail(EB_EINVAL, "eb_mlp_set_obs_scale: null handle");
    m->has_scale = scale != nullptr;
    if (scale) {
        EB_HIP(hipSetDevice(m->cfg.device));
        EB_HIP(hipMemcpy(m->d_scale, scale, sizeof(float) * m->cfg.obs_dim, hipMemcpyHostToDevice));
    }
    return EB_OK;
}

int eb_mlp_forward(eb_mlp m, int32_t n, const float* obs, float* out, void* stream) {
    eb::MlpArgs A;
    int rc = mlp_args(m, n, obs, out, eb::MLP_HEAD_LOGITS, 0.0f, &A, "eb_mlp_forward: null handle");
    if (rc || n == 0) return rc;
    EB_HIP(hipSetDevice(m->cfg.device));
    EB_HIP(eb::launch_mlp(A, (hipStream_t)stream));
    return EB_OK;
}

int eb_policy_run_batch(eb_mlp m, int32_t n, const float* obs, float action_range, float* actions, void* stream) {
    eb::MlpArgs A;
    int rc = mlp_args(m, n, obs, actions, eb::MLP_HEAD_ACTION, action_range, &A, "eb_policy_run_batch: null handle");
    if (rc || n == 0) return rc;
    EB_HIP(hipSetDevice(m->cfg.device));
    EB_HIP(eb::launch_mlp(A, (hipStream_t)stream));
    return EB_OK;
}

int eb_shield_is_safe(eb_handle h, eb_mlp policy, int32_t n_env, const float* obs_in, const int32_t* ref_idx,
                      int32_t path_id, int32_t steps, int32_t penalty, float action_range, float* obs_a,
                      float* obs_b, float* actions, float* out5, float* punish, uint8_t* safe, void* stream) {
    if (h && policy && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_shield_is_safe: null handle");
    if (rc) return rc;
    if (!policy) return fail(EB_EINVAL, "eb_shield_is_safe: null policy");
    if (n_env < 0 || steps < 1 || !obs_in || !obs_a || !obs_b || !actions || !out5 || !punish || !safe || obs_a == obs_b ||
        obs_in == obs_a || obs_in == obs_b)
        return fail(EB_EINVAL, "eb_shield_is_safe: bad argument");
    if (penalty != EB_PENALTY_VEH2VEH4REAL && penalty != EB_PENALTY_REAL_PUNISH_TERM) return fail(EB_EINVAL, "eb_shield_is_safe: unknown penalty");
    if (policy->cfg.obs_dim != obs_dim(h->cfg) || policy->cfg.out_dim != 4) return fail(EB_EINVAL, "eb_shield_is_safe: the policy does not fit the model (obs_dim, out_dim = 4)");
    if (policy->cfg.device != h->cfg.device) return fail(EB_EINVAL, "eb_shield_is_safe: policy and model live on different devices");
    EB_HIP(hipSetDevice(h->cfg.device));
    hipStream_t s = pick(h, stream);
    eb::MlpArgs A;
    // punish += penalty (hier_decision.py:93-97) rides on the step's launch: the env wave that has just made out5's rows 2 / 3 adds
    // the one asked for to the running sum and, in the last look-ahead, sets the flag — two launches per look-ahead instead of three
    const float* cur = obs_in;
    for (int t = 0; t < steps; ++t) {
        float* dst = (t & 1) ? obs_b : obs_a;
        rc = mlp_args(policy, n_env, cur, actions, eb::MLP_HEAD_ACTION, action_range, &A, "eb_shield_is_safe: null policy");
        if (rc) return rc;
        EB_HIP(eb::launch_mlp(A, s));
        const ShieldArgs sh{punish, safe, penalty == EB_PENALTY_VEH2VEH4REAL ? 3 : 2, t == 0, t == steps - 1};   // rows of rollout_out's outputs (DAM:126)
        rc = rollout_common(h, n_env, cur, actions, ref_idx, path_id, dst, out5, nullptr, 1, 1, s, 0, 0, nullptr, nullptr, &sh);
        if (rc) return rc;
        cur = dst;
    }
    return EB_OK;
}

}  // extern "C"
