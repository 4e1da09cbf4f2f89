// index.hip — C ABI (include/bergen_hip.h): resident flat index + exact search orchestration.
//
// Reference behaviour being replaced (naver/bergen):
//   modules/retrieve.py:84-90    all chunk files loaded to HOST RAM            -> one HBM-resident index
//   modules/retrieve.py:152-164  per query chunk: H2D of every doc chunk, torch.mm, torch.topk
//                                                                              -> bh_scan_topk_kernel
//   modules/retrieve.py:165-166  IOError when the index has fewer rows than the dataset
//                                                                              -> BH_EINCOMPLETE
//   modules/retrieve.py:169-177  host concat + fp32 topk + gather              -> bh_merge_rescore_kernel
// There is no CPU fallback here: every entry point that computes needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/bergen_hip.h"
#include "bh_host.h"
#include "bh_kernels.h"

namespace {
thread_local std::string g_err;
#define fail(...) bh_fail(__VA_ARGS__)

#define HIP_TRY(expr) BH_HIP_TRY(expr)

}  // namespace

// shared by every translation unit of the library (bh_kernels.h)
int bh_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

struct Options {
    int query_tile = 128;
    int share_threshold = 1;
    int nontemporal = 1;
    int workgroups_per_cu = 1;
    int ablate = 0;
    int ring_variant = 0;
    int query_split = 1;
    int pair_window = 0;
    int dma_interleave = 1;
    int dyn_tiles = 1;  // scan_topk256: the last eighth of the corpus in chunks by claim counter (0: everything round robin)
    // 3 = scan_topk256.hip (8 waves, 256-query tile) where it applies (d in {384, 512, 768}), else scan_topk.hip;
    // 2 = scan_topk192.hip where it applies (d = 768, k <= 56); 0 = scan_topk.hip (4 waves, 128-query tile)
    int scan_kernel = 3;
    int certify = 1;  // exactness certificate + exact fall-back scan for the queries it cannot prove (certify.hip)
    int tail128 = 1;  // scan_topk256: a last pass of at most 128 queries runs on the 128-query kernel (5.6 instead of 7.2 ms at 21 M x 768)
    int err_scale = 1;  // test-only: multiplies the certificate's error bound (forces queries through the fall-back)
    int filter256 = 1;  // exact fall-back: filter passes of 256 queries on scan_topk256.hip where it applies (0: 128 queries on scan_topk.hip)
    int balance_tail = 1;  // scan_topk256 + pair256: the remainder of a query set (more than one tile, fewer than two) as ONE balanced paired launch (search_device_lists)
    int pair256 = 1;    // scan_topk256: workgroups b and b ^ 8 (same XCD) walk the same tiles for two 256-query passes of one launch (1 = paced every 16 tiles, 2 = free-running, 3 = paced with the non-temporal stream policy of the unpaired launches)
};

// One row per dense-search option: name, member, accepted values (lo..hi, or a short list), the message of a rejected value.
struct OptionDef {
    const char* name;
    int Options::*member;
    long long lo, hi;
    long long only[4];  // when only[0] != -1: the accepted values (terminated by -1)
    const char* expect;
};
const OptionDef g_option_defs[] = {
    {"query_tile", &Options::query_tile, 0, 0, {128, 256, -1, -1}, "query_tile must be 128 or 256"},
    {"share_threshold", &Options::share_threshold, 0, 1, {-1, -1, -1, -1}, "share_threshold must be 0 or 1"},
    {"nontemporal", &Options::nontemporal, 0, 1, {-1, -1, -1, -1}, "nontemporal must be 0 or 1"},
    {"ablate", &Options::ablate, 0, 63, {-1, -1, -1, -1}, "ablate must be 0..63"},  // bench-only: results are NOT valid search results when != 0
    {"query_split", &Options::query_split, 1, 2, {-1, -1, -1, -1}, "query_split must be 1 or 2"},
    {"dma_interleave", &Options::dma_interleave, 0, 1, {-1, -1, -1, -1}, "dma_interleave must be 0 or 1"},
    {"scan_kernel", &Options::scan_kernel, 0, 0, {0, 2, 3, -1},
     "scan_kernel must be 0 (128-query tile), 2 (192-query tile) or 3 (256-query tile, two waves per SIMD)"},
    {"certify", &Options::certify, 0, 1, {-1, -1, -1, -1}, "certify must be 0 or 1"},
    {"tail128", &Options::tail128, 0, 1, {-1, -1, -1, -1}, "tail128 must be 0 or 1"},
    // (results stay exact: a looser bound only sends more queries through the fall-back)
    {"certificate_error_scale", &Options::err_scale, 1, 1 << 24, {-1, -1, -1, -1}, "certificate_error_scale must be 1..2^24"},
    {"dyn_tiles", &Options::dyn_tiles, 0, 1, {-1, -1, -1, -1}, "dyn_tiles must be 0 or 1"},
    {"pair_window", &Options::pair_window, 0, 64, {-1, -1, -1, -1}, "pair_window must be 0..64"},
    {"ring_variant", &Options::ring_variant, 0, 7, {-1, -1, -1, -1}, "ring_variant must be 0..7"},
    {"workgroups_per_cu", &Options::workgroups_per_cu, 1, 1, {-1, -1, -1, -1}, "workgroups_per_cu must be 1 (LDS ring fills the CU)"},
    {"filter256", &Options::filter256, 0, 1, {-1, -1, -1, -1}, "filter256 must be 0 or 1"},
    {"pair256", &Options::pair256, 0, 3, {-1, -1, -1, -1}, "pair256 must be 0..3"},
    {"balance_tail", &Options::balance_tail, 0, 1, {-1, -1, -1, -1}, "balance_tail must be 0 or 1"},
};
constexpr int kUnset = INT32_MIN;  // per-handle override table: "inherit the process-wide value"

// The process-wide defaults (bh_set_option).  Plain ints written by one call and read ONCE per search into a snapshot
// (effective_options): a concurrent toggle cannot change a running search, and a handle's own overrides
// (bh_index_set_option) are not visible to any other handle.
Options g_opt;

const OptionDef* find_option(const char* name) {
    for (const OptionDef& d : g_option_defs)
        if (strcmp(d.name, name) == 0) return &d;
    return nullptr;
}
bool option_accepts(const OptionDef& d, long long v) {
    if (d.only[0] != -1) {
        for (long long o : d.only)
            if (o != -1 && o == v) return true;
        return false;
    }
    return v >= d.lo && v <= d.hi;
}

int pad_dim(int dim) {
    static const int sizes[] = {64, 128, 256, 384, 512, 768, 1024};
    for (int s : sizes)
        if (dim <= s) return s;
    return -1;
}

int pick_kp(int k) {
    if (k <= 56) return 64;
    if (k <= 120) return 128;
    if (k <= 248) return 256;
    return -1;
}

template <typename T>
using DevBuf = BhDevBuf<T>;

}  // namespace

struct bh_index {
    int device = 0;
    int n_cu = 256;
    int64_t n_rows = 0;
    int64_t n_tiles = 0;
    int dim = 0, dim_padded = 0;
    int metric = 0;
    _Float16* rows = nullptr;
    bool finalized = false;
    std::vector<std::pair<int64_t, int64_t>> have;  // merged [begin, end) intervals uploaded
    hipStream_t stream = nullptr;
    hipStream_t merge_stream = nullptr;  // merge / re-score of pass p runs beside the scan of pass p + 1
    DevBuf<bh_u64> cand, partial;
    DevBuf<unsigned> gthr;
    DevBuf<bh_u64> clk;  // [grid][8] phase stamps of the last scan launch + the timeline words (diagnostics, scan_topk256.hip)
    float max_norm = 0.f;        // largest row norm (after normalisation for cosine), set by bh_index_finalize
    DevBuf<unsigned> uncert;     // [nq] certificate flags of the last search
    unsigned* n_uncert_host = nullptr;  // host-mapped count of uncertified queries of the running search
    unsigned* n_uncert_dev = nullptr;
    DevBuf<bh_u64> kth;          // [nq] canonical key of each query's k-th result
    DevBuf<_Float16> exact_q;    // [BH_EXACT_BATCH][D] uncertified queries gathered for the exact scan
    DevBuf<bh_u64> exact_keys;   // [BH_EXACT_BATCH][BH_EXACT_CAP] + [BH_EXACT_BATCH] thresholds
    DevBuf<unsigned> exact_cnt;  // [4][BH_EXACT_BATCH] counts | thresholds | query indices
    DevBuf<unsigned> exact_rows; // [BH_EXACT_BATCH][BH_EXACT_CAP] rows the filter pass let through
    DevBuf<_Float16> qbuf;
    DevBuf<unsigned char> staging;
    unsigned char* pinned[2] = {nullptr, nullptr};  // host staging of the load path (upload_common)
    size_t pinned_cap[2] = {0, 0};
    hipEvent_t pinned_ev[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> events;
    bh_counters counters{};
    int last_shape_nq = -1, last_shape_k = -1;  // the previous search's shape (queries, k, rows searched: search_large_k runs range views) and
    int64_t last_shape_rows = -1;                // GPU time: when to stop sleeping between polls (spin_sync)
    double last_shape_ms = 0.0;
    int opt_override[sizeof(g_option_defs) / sizeof(g_option_defs[0])];  // kUnset = inherit (bh_index_set_option)

    bh_index() {
        for (int& o : opt_override) o = kUnset;
    }

    hipEvent_t event(size_t i) {
        while (events.size() <= i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            events.push_back(e);
        }
        return events[i];
    }
};

namespace {

Options effective_options(const bh_index* ix) {
    Options o = g_opt;
    for (size_t j = 0; j < sizeof(g_option_defs) / sizeof(g_option_defs[0]); ++j)
        if (ix->opt_override[j] != kUnset) o.*(g_option_defs[j].member) = ix->opt_override[j];
    return o;
}

// Wait for a stream by polling it.  hipStreamSynchronize parks the thread once the wait gets long (tens of milliseconds:
// every search over a whole corpus) and the wake-up costs 1-2 ms, 2 % of the headline search.  A full-speed spin for the
// whole search would hold a host core per rank (8 ranks per node next to tokenizer workers), so: spin for the first
// 200 us (short searches), then poll every 50 us with the thread asleep in between (wake-up latency <= ~0.1 ms on an idle host, a
// few per cent of a core) — between 80 % and 130 % of `expect_ms` (the duration of the handle's previous search of this shape over
// this many rows) the thread spins without sleeping.  On a host loaded by other tenants a 50 us sleep can come back milliseconds late
// (round 4: a 97 ms step around 87 ms of kernels on a box at load average 100), and only the last sleep of a search costs
// anything.  The window is bounded: a search that runs LONGER than the last one (a fall-back pass, the GPU shared with an encoder
// or with other ranks) goes back to 50 us sleeps past 130 % instead of holding a host core for up to 10 s — eight ranks spinning
// under a 16-CPU quota are the CFS throttling utils.cpu_budget exists to avoid.  The blocking wait takes over after 10 s.
hipError_t spin_sync(hipStream_t st, double expect_ms = 0.0) {
    const auto t0 = std::chrono::steady_clock::now();
    const auto spin_from = std::chrono::microseconds((long long)(expect_ms * 800.0));
    const auto spin_until = std::chrono::microseconds((long long)(expect_ms * 1300.0));
    for (unsigned n = 0;; ++n) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((n & 15u) != 15u) continue;
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::seconds(10)) return hipStreamSynchronize(st);
        if (waited > std::chrono::microseconds(200) && (expect_ms <= 0.0 || waited < spin_from || waited > spin_until))
            std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// a host-to-host copy on a few threads (one thread moves ~10 GB/s; the PCIe link takes five times that)
void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = bytes < (8u << 20) ? 1 : (int)std::min<unsigned>(8u, std::max<unsigned>(1u, hw / 2));
    if (nt == 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (int t = 0; t < nt; ++t) {
        const size_t lo = std::min(bytes, (size_t)t * per), hi = std::min(bytes, lo + per);
        if (hi > lo) th.emplace_back([=] { memcpy((unsigned char*)dst + lo, (const unsigned char*)src + lo, hi - lo); });
    }
    for (auto& t : th) t.join();
}

void add_interval(std::vector<std::pair<int64_t, int64_t>>& v, int64_t b, int64_t e) {
    v.emplace_back(b, e);
    std::sort(v.begin(), v.end());
    std::vector<std::pair<int64_t, int64_t>> out;
    for (auto& iv : v) {
        if (!out.empty() && iv.first <= out.back().second)
            out.back().second = std::max(out.back().second, iv.second);
        else
            out.push_back(iv);
    }
    v.swap(out);
}

int64_t rows_have(const bh_index* ix) {
    int64_t n = 0;
    for (auto& iv : ix->have) n += iv.second - iv.first;
    return n;
}

int upload_common(bh_index* ix, int64_t row0, const void* src, int64_t n, int32_t src_dtype, bool src_on_device) {
    if (!ix) return fail(BH_EINVAL, "null index");
    if (n < 0 || row0 < 0 || row0 + n > ix->n_rows)
        return fail(BH_EINVAL, "rows [%lld, %lld) outside the index (n_rows=%lld)", (long long)row0,
                    (long long)(row0 + n), (long long)ix->n_rows);
    if (src_dtype != BH_F16 && src_dtype != BH_F32) return fail(BH_EINVAL, "bad src_dtype %d", src_dtype);
    if (n == 0) return BH_OK;
    if (!src) return fail(BH_EINVAL, "null source");
    if (ix->finalized && ix->metric == BH_METRIC_COS)
        return fail(BH_EINVAL, "cosine index already finalized (rows are normalised in place)");
    HIP_TRY(hipSetDevice(ix->device));
    const size_t esz = src_dtype == BH_F16 ? 2 : 4;
    _Float16* dst = ix->rows + (size_t)row0 * ix->dim_padded;
    const bool direct = src_dtype == BH_F16 && ix->dim == ix->dim_padded;  // the bytes land as they are
    if (src_on_device) {
        if (direct)
            HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * ix->dim * 2, hipMemcpyDeviceToDevice, ix->stream));
        else
            HIP_TRY(bh_launch_convert_rows(src, src_dtype, n, ix->dim, dst, ix->dim_padded, ix->stream));
    } else {
        // Host source (a chunk file torch.load-ed or mmap-ed into pageable memory: the index load path, reference
        // modules/retrieve.py:153).  A pageable hipMemcpy stages through the runtime's own small pinned buffers at a few
        // GB/s; here the block goes through TWO pinned buffers of this index: a few host threads copy piece i + 1 into
        // one while the DMA engine moves piece i out of the other (and, for fp32 or padded rows, the conversion kernel
        // reads it from a device staging area).
        const size_t row_bytes = (size_t)ix->dim * esz;
        const size_t piece_bytes = 64ull << 20;
        const int64_t rows_per = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(piece_bytes / row_bytes)));
        for (int b = 0; b < 2; ++b) {
            if (ix->pinned_cap[b] < (size_t)rows_per * row_bytes) {
                if (ix->pinned[b]) (void)hipHostFree(ix->pinned[b]);
                ix->pinned[b] = nullptr;
                ix->pinned_cap[b] = 0;
                HIP_TRY(hipHostMalloc((void**)&ix->pinned[b], (size_t)rows_per * row_bytes, hipHostMallocDefault));
                ix->pinned_cap[b] = (size_t)rows_per * row_bytes;
            }
            if (!ix->pinned_ev[b]) HIP_TRY(hipEventCreateWithFlags(&ix->pinned_ev[b], hipEventDisableTiming));
        }
        if (!direct) {
            int rc = ix->staging.ensure(2 * (size_t)rows_per * row_bytes);
            if (rc) return rc;
        }
        bool used[2] = {false, false};
        int64_t piece = 0;
        for (int64_t r = 0; r < n; r += rows_per, ++piece) {
            const int b = (int)(piece & 1);
            const int64_t m = std::min<int64_t>(rows_per, n - r);
            const size_t bytes = (size_t)m * row_bytes;
            if (used[b]) HIP_TRY(hipEventSynchronize(ix->pinned_ev[b]));  // the DMA out of this buffer has finished
            parallel_memcpy(ix->pinned[b], (const unsigned char*)src + (size_t)r * row_bytes, bytes);
            if (direct) {
                HIP_TRY(hipMemcpyAsync(dst + (size_t)r * ix->dim_padded, ix->pinned[b], bytes, hipMemcpyHostToDevice, ix->stream));
            } else {
                unsigned char* stage = ix->staging.p + (size_t)b * (size_t)rows_per * row_bytes;
                HIP_TRY(hipMemcpyAsync(stage, ix->pinned[b], bytes, hipMemcpyHostToDevice, ix->stream));
                HIP_TRY(bh_launch_convert_rows(stage, src_dtype, m, ix->dim, dst + (size_t)r * ix->dim_padded, ix->dim_padded, ix->stream));
            }
            HIP_TRY(hipEventRecord(ix->pinned_ev[b], ix->stream));
            used[b] = true;
        }
    }
    HIP_TRY(hipStreamSynchronize(ix->stream));
    add_interval(ix->have, row0, row0 + n);
    return BH_OK;
}

}  // namespace

extern "C" {

int bh_version(void) { return BH_VERSION; }
const char* bh_last_error(void) { return g_err.c_str(); }

int bh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bh_init(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(BH_EHIP, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(BH_EINVAL, "device %d out of range [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BH_EUNSUPPORTED, "device %d is %s; this library contains gfx950 (MI355X) code only", device_id,
                    prop.gcnArchName);
    return BH_OK;
}

int bh_set_option(const char* name, int64_t value) {
    if (!name) return fail(BH_EINVAL, "null option name");
    std::string s(name);
    if (const OptionDef* d = find_option(name)) {
        if (!option_accepts(*d, value)) return fail(BH_EINVAL, "%s", d->expect);
        g_opt.*(d->member) = (int)value;
    } else if (s == "sparse_kernel") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "sparse_kernel must be 0 (broadcast) or 1 (mfma)");
        bh_sparse_set_kernel((int)value);
    } else if (s == "sparse_head") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "sparse_head must be 0 (plain CSR stream) or 1 (corpus-head tiles + tail stream)");
        bh_sparse_set_head((int)value);
    } else if (s == "sparse_ablate") {
        if (value < 0 || value > 2047) return fail(BH_EINVAL, "sparse_ablate must be 0..2047");
        bh_sparse_set_ablate((int)value);
    } else if (s == "gemm_stagger_phases") {
        if (value < 0 || value > 64) return fail(BH_EINVAL, "gemm_stagger_phases must be 0..64");
        bh_gemm_set_stagger((int)value, -1);
    } else if (s == "gemm_full_line_stores") {
        if (value < 0 || value > 2) return fail(BH_EINVAL, "gemm_full_line_stores must be 0, 1 or 2 (1 = row-major outputs, 2 = the default: + blocked V^T output and gated fold)");
        bh_gemm_set_full_line_stores((int)value);
    } else if (s == "gemm_mfma16") {
        if (value < 0 || value > 511 || (value > 1 && (value & 15) != 1)) return fail(BH_EINVAL, "gemm_mfma16 must be 0, 1 or 16 x ablation bits + 1");
        bh_gemm_set_mfma16((int)value);
    } else if (s == "gemm_rotary_fused") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "gemm_rotary_fused must be 0 or 1");
        bh_gemm_set_rotary_fused((int)value);
    } else if (s == "ln_small") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "ln_small must be 0 or 1");
        bh_ln_set_small((int)value);
    } else if (s == "gemm_tail_split") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "gemm_tail_split must be 0 or 1");
        bh_gemm_set_tail_split((int)value);
    } else if (s == "gemm_gelu_nontemporal") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "gemm_gelu_nontemporal must be 0 or 1");
        bh_gemm_set_gelu_nontemporal((int)value);
    } else if (s == "attention_rel_wide_stores") {
        if (value != 0 && value != 1) return fail(BH_EINVAL, "attention_rel_wide_stores must be 0 or 1");
        bh_attention_rel_set_wide_stores((int)value);
    } else if (s == "gemm_stagger_pct") {
        if (value < 1 || value > 400) return fail(BH_EINVAL, "gemm_stagger_pct must be 1..400");
        bh_gemm_set_stagger(-1, (int)value);
    } else {
        return fail(BH_EINVAL, "unknown option '%s'", name);
    }
    return BH_OK;
}

int bh_index_set_option(bh_index* ix, const char* name, int64_t value) {
    if (!ix) return fail(BH_EINVAL, "null index");
    if (!name) return fail(BH_EINVAL, "null option name");
    const OptionDef* d = find_option(name);
    if (!d) return fail(BH_EINVAL, "unknown per-index option '%s'", name);
    const size_t slot = (size_t)(d - g_option_defs);
    if (value == BH_OPTION_INHERIT) {
        ix->opt_override[slot] = kUnset;
        return BH_OK;
    }
    if (!option_accepts(*d, value)) return fail(BH_EINVAL, "%s", d->expect);
    ix->opt_override[slot] = (int)value;
    return BH_OK;
}

int bh_index_create(bh_index** out, int64_t n_rows, int32_t dim, int32_t dtype, int32_t metric) {
    if (!out) return fail(BH_EINVAL, "null out");
    *out = nullptr;
    if (n_rows < 0 || n_rows >= 0xffffffffll) return fail(BH_EINVAL, "n_rows %lld out of range", (long long)n_rows);
    if (dim <= 0) return fail(BH_EINVAL, "dim must be positive");
    if (dtype != BH_F16)
        return fail(BH_EUNSUPPORTED, "index storage dtype %d unsupported (fp16 only; fp32 sources are rounded "
                                     "at upload like the reference's fp16 embeddings)", dtype);
    if (metric != BH_METRIC_IP && metric != BH_METRIC_COS) return fail(BH_EINVAL, "bad metric %d", metric);
    const int dp = pad_dim(dim);
    if (dp < 0) return fail(BH_EUNSUPPORTED, "dim %d unsupported (max 1024)", dim);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BH_EUNSUPPORTED, "device %d is %s; gfx950 required", dev, prop.gcnArchName);
    bh_index* ix = new bh_index();
    ix->device = dev;
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ix->n_rows = n_rows;
    ix->n_tiles = (n_rows + 31) / 32;
    ix->dim = dim;
    ix->dim_padded = dp;
    ix->metric = metric;
    const size_t bytes = std::max<size_t>((size_t)ix->n_tiles * 32 * dp * 2, 64);
    hipError_t e = hipMalloc((void**)&ix->rows, bytes);
    if (e != hipSuccess) {
        delete ix;
        return fail(e == hipErrorOutOfMemory ? BH_ENOMEM : BH_EHIP, "hipMalloc(%zu bytes) for the index: %s", bytes,
                    hipGetErrorString(e));
    }
    e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipFree(ix->rows);
        delete ix;
        return fail(BH_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    // zero the row padding of the last tile (and the column padding when the source is dense fp16)
    if (n_rows % 32 != 0 || dim != dp) {
        const size_t tail_from = (dim != dp) ? 0 : (size_t)n_rows * dp * 2;
        e = hipMemsetAsync((unsigned char*)ix->rows + tail_from, 0, bytes - tail_from, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e != hipSuccess) {
            bh_index_destroy(ix);
            return fail(BH_EHIP, "memset: %s", hipGetErrorString(e));
        }
    }
    *out = ix;
    return BH_OK;
}

void bh_index_destroy(bh_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    if (ix->merge_stream) (void)hipStreamDestroy(ix->merge_stream);
    for (auto e : ix->events) (void)hipEventDestroy(e);
    ix->cand.release();
    ix->partial.release();
    ix->gthr.release();
    ix->clk.release();
    ix->uncert.release();
    if (ix->n_uncert_host) (void)hipHostFree(ix->n_uncert_host);
    ix->n_uncert_host = ix->n_uncert_dev = nullptr;
    ix->kth.release();
    ix->exact_q.release();
    ix->exact_keys.release();
    ix->exact_cnt.release();
    ix->exact_rows.release();
    ix->qbuf.release();
    ix->staging.release();
    for (int b = 0; b < 2; ++b) {
        if (ix->pinned[b]) (void)hipHostFree(ix->pinned[b]);
        if (ix->pinned_ev[b]) (void)hipEventDestroy(ix->pinned_ev[b]);
        ix->pinned[b] = nullptr;
        ix->pinned_ev[b] = nullptr;
    }
    if (ix->rows) (void)hipFree(ix->rows);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

int bh_index_upload(bh_index* ix, int64_t row0, const void* host_rows, int64_t n, int32_t src_dtype) {
    return upload_common(ix, row0, host_rows, n, src_dtype, false);
}
int bh_index_upload_device(bh_index* ix, int64_t row0, const void* dev_rows, int64_t n, int32_t src_dtype) {
    return upload_common(ix, row0, dev_rows, n, src_dtype, true);
}

int64_t bh_index_rows_uploaded(const bh_index* ix) { return ix ? rows_have(ix) : 0; }

int bh_index_finalize(bh_index* ix) {
    if (!ix) return fail(BH_EINVAL, "null index");
    const int64_t have = rows_have(ix);
    if (have != ix->n_rows)
        return fail(BH_EINCOMPLETE, "!!! Index is not complete. Please re-index. Missing %lld documents in the index. !!!",
                    (long long)(ix->n_rows - have));
    if (ix->finalized) return BH_OK;
    HIP_TRY(hipSetDevice(ix->device));
    if (ix->metric == BH_METRIC_COS) {
        HIP_TRY(bh_launch_l2_normalize_rows(ix->rows, ix->n_rows, ix->dim, ix->dim_padded, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    // largest row norm: the scale of the exactness certificate's error bound (certify.hip); one pass over the corpus
    ix->max_norm = 0.f;
    if (ix->n_rows > 0) {
        int rc = ix->exact_cnt.ensure(4 * BH_EXACT_BATCH);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(ix->exact_cnt.p, 0, sizeof(unsigned), ix->stream));
        HIP_TRY(bh_launch_row_norm_max(ix->rows, ix->n_rows, ix->dim_padded, ix->exact_cnt.p, ix->stream));
        unsigned bits = 0;
        HIP_TRY(hipMemcpyAsync(&bits, ix->exact_cnt.p, sizeof(unsigned), hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
        float n2;
        memcpy(&n2, &bits, sizeof n2);
        ix->max_norm = std::sqrt(n2);
    }
    ix->finalized = true;
    return BH_OK;
}

namespace {
int search_large_k(bh_index* ix, const void* q_dev, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                   float* out_scores_dev, int64_t* out_ids_dev);
}

// k <= 248: one fused search (scan + merge / re-score + certificate).  Larger k: search_large_k below.
static int search_device_lists(bh_index* ix, const void* q_dev, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                               float* out_scores_dev, int64_t* out_ids_dev);

int bh_search_device(bh_index* ix, const void* q_dev, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                     float* out_scores_dev, int64_t* out_ids_dev) {
    if (!ix) return fail(BH_EINVAL, "null index");
    if (k > BH_MAX_LIST_K && k <= BH_MAX_K && nq > 0 && ix->finalized)
        return search_large_k(ix, q_dev, q_dtype, nq, k, id_offset, out_scores_dev, out_ids_dev);
    return search_device_lists(ix, q_dev, q_dtype, nq, k, id_offset, out_scores_dev, out_ids_dev);
}

static int search_device_lists(bh_index* ix, const void* q_dev, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                               float* out_scores_dev, int64_t* out_ids_dev) {
    if (!ix) return fail(BH_EINVAL, "null index");
    if (!ix->finalized) {
        const int64_t have = rows_have(ix);
        if (have != ix->n_rows)
            return fail(BH_EINCOMPLETE,
                        "!!! Index is not complete. Please re-index. Missing %lld documents in the index. !!!",
                        (long long)(ix->n_rows - have));
        return fail(BH_EINVAL, "index not finalized");
    }
    if (nq < 0 || k <= 0) return fail(BH_EINVAL, "nq=%d k=%d", nq, k);
    if (q_dtype != BH_F16 && q_dtype != BH_F32) return fail(BH_EINVAL, "bad q_dtype %d", q_dtype);
    const int kp = pick_kp(k);
    if (kp < 0) return fail(BH_EUNSUPPORTED, "k=%d unsupported (max %d)", k, BH_MAX_K);
    if (nq == 0) return BH_OK;
    if (!q_dev || !out_scores_dev || !out_ids_dev) return fail(BH_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(ix->device));

    const Options opt = effective_options(ix);  // one snapshot per search: process-wide values + this handle's overrides
    const int dp = ix->dim_padded;
    int qw = (opt.query_tile == 256 && bh_scan_supports(dp, kp, 2)) ? 2 : 1;
    const bool use256 = opt.scan_kernel == 3 && bh_scan256_supports(dp, kp);
    const bool use192 = opt.scan_kernel == 2 && bh_scan192_supports(dp, kp);
    if (use192 || use256) qw = 1;
    const int bq = use256 ? bh_scan256_tile(dp) : use192 ? 192 : 128 * qw;
    const int grid = ix->n_cu * opt.workgroups_per_cu;
    // passes: a launch scans for qs * bq queries (qs = 2: paired workgroups share the corpus stream through L2,
    // scan_topk.hip); the last queries run unsplit when no more than bq are left
    const int qs_max = (opt.query_split == 2 && grid % 16 == 0 && !use192 && !use256) ? 2 : 1;
    std::vector<std::pair<int, int>> passes;  // (first query, qs)
    for (int q0 = 0; q0 < nq;) {
        const int qs = (qs_max == 2 && nq - q0 > bq) ? 2 : 1;
        passes.push_back({q0, qs});
        q0 += bq * qs;
    }
    // BALANCED REMAINDER (option balance_tail, default on; scan_topk256 with paired launches): when the queries left behind the last
    // full pair of passes are more than one tile (bq < R < 2 bq — 2 837 = 5 x 512 + 277), they used to run as a full pass alone on the
    // whole chip plus a tail pass on the 128-query kernel (7.1 + 4.9 ms, the tail another corpus read for 21 queries).  Now they are
    // cut into two passes of about R / 2 queries — the first a whole number of 32-query waves — that run as ONE more paired launch:
    // the corpus is read once for both, and the waves of a workgroup that hold no query at all sit the tiles out (BhScanArgs::
    // nq_valid: no fragment reads, no MFMAs, no filter), so each half-grid carries half the matrix work of a full pass.  The second
    // pass's query tile starts where the first one's queries end (BhScanArgs::qtile2): the query buffer stays contiguous.
    int bal_first = -1, bal_nq[2] = {0, 0};  // pass index of the balanced pair (its first pass) and the two passes' query counts
    if (use256 && qs_max == 1 && opt.balance_tail != 0 && opt.pair256 != 0 && opt.ablate == 0 && grid == 256 && (int)passes.size() >= 2 &&
        bh_scan256_pair_supports(dp, kp) && (dp != 1024 || opt.ring_variant == 0 || opt.ring_variant == 5)) {
        const int np = (int)passes.size();
        const int n_full_pairs = (np - 2) / 2 * 2 == np - 2 ? (np - 2) / 2 : -1;  // the last two passes must be a pair of their own
        const int rem = nq - (np - 2) * bq;                                        // queries of the last two passes: bq < rem <= 2 bq
        if (n_full_pairs >= 0 && rem > bq && rem < 2 * bq) {
            const int per_wave = bq / 8;
            bal_nq[0] = ((rem + 1) / 2 + per_wave - 1) / per_wave * per_wave;      // whole waves: the tile behind them belongs to the next pass
            if (bal_nq[0] > bq) bal_nq[0] = bq;
            bal_nq[1] = rem - bal_nq[0];
            bal_first = np - 2;
            passes[(size_t)np - 1].first = passes[(size_t)np - 2].first + bal_nq[0];
        }
    }
    const int n_pass = (int)passes.size();
    const int64_t nq_pad = (int64_t)((nq + bq - 1) / bq) * bq;

    int rc;
    if ((rc = ix->qbuf.ensure((size_t)nq_pad * dp))) return rc;
    if ((rc = ix->cand.ensure((size_t)grid * bq * 2 * kp))) return rc;
    const size_t partial_elems = (size_t)grid * bq * kp;
    // Unsplit passes (every kernel but scan_topk.hip's paired workgroups) are merged in GROUPS: the scans of a group run
    // back to back, then ONE merge launch takes all their queries.  A scan workgroup fills the register file of its CU, so
    // a merge beside the next pass's scan only ran in the gaps (it delayed the scan's workgroups by its ~60 us, or ran
    // after them and held up the pass that reuses its lists: 10 % of a pass over an eighth of the corpus), and one
    // workgroup per query of ONE pass is latency-bound; thousands of them at once are not.  Up to 1 GiB of list sets.
    const bool grouped = qs_max == 1;
    const int group = grouped ? (int)std::max<size_t>(1, std::min<size_t>((size_t)n_pass, ((size_t)1 << 30) / (partial_elems * sizeof(bh_u64)))) : 2;
    if ((rc = ix->partial.ensure((size_t)group * partial_elems))) return rc;  // (paired: two sets, pass p is merged while p + 1 is scanned)
    // one threshold block per pass (scan_topk256: [bq][256] slots + [bq] bounds + the claim counter; else [bq * qs][64] slots):
    // all of them are reset by ONE launch up front; the paired-workgroup progress words sit behind the blocks
    const size_t gthr_pass = use256 ? (size_t)bq * (BH_SLOTS256 + 1) + 4 : (size_t)bq * qs_max * 64;
    // (the progress words of paired workgroups sit behind the blocks, 64-byte aligned: partners are 32 bytes apart)
    const size_t progress_off = (gthr_pass * (size_t)n_pass + 15) / 16 * 16;
    if ((rc = ix->gthr.ensure(progress_off + grid))) return rc;
    if ((rc = ix->clk.ensure((size_t)grid * 8 + BH_TL_WORDS, true, ix->stream))) return rc;
    if ((rc = ix->uncert.ensure((size_t)nq_pad))) return rc;
    if ((rc = ix->kth.ensure((size_t)nq_pad))) return rc;
    if (!ix->n_uncert_host) {  // host-mapped: the merge kernel counts the queries it could not certify straight into host memory
        HIP_TRY(hipHostMalloc((void**)&ix->n_uncert_host, sizeof(unsigned), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&ix->n_uncert_dev, ix->n_uncert_host, 0));
    }
    *ix->n_uncert_host = 0u;
    // |mfma - canonical| <= 2 d 2^-24 |q| |x| for any summation order of d exact products in fp32
    const float err_coef = opt.certify ? 2.0f * (float)dp * 5.9604645e-8f * ix->max_norm * (float)opt.err_scale : 0.f;

    hipStream_t st = ix->stream;
    if (!ix->merge_stream) HIP_TRY(hipStreamCreateWithFlags(&ix->merge_stream, hipStreamNonBlocking));
    hipStream_t ms = ix->merge_stream;
    // queries -> padded fp16 tile buffer (zero rows beyond nq)
    if (nq_pad > nq) HIP_TRY(hipMemsetAsync(ix->qbuf.p + (size_t)nq * dp, 0, (size_t)(nq_pad - nq) * dp * 2, st));
    HIP_TRY(bh_launch_convert_rows(q_dev, q_dtype, nq, ix->dim, ix->qbuf.p, dp, st));
    if (ix->metric == BH_METRIC_COS) HIP_TRY(bh_launch_l2_normalize_rows(ix->qbuf.p, nq, ix->dim, dp, st));

    hipEvent_t ev_begin = ix->event(0), ev_end = ix->event(1);
    if (!ev_begin || !ev_end) return fail(BH_EHIP, "hipEventCreate failed");
    // scan_topk256: a last pass of at most 128 queries (2 837 = 11 x 256 + 21) runs on the 128-query kernel — the same
    // corpus pass costs 5.6 instead of 7.2 ms there —, as a group of its own (its lists are 128 queries wide)
    const bool tail128 = grouped && use256 && bq == 256 && opt.tail128 && opt.ablate == 0 && nq % bq != 0 && nq % bq <= 128 && bal_first < 0;
    const int n_main = tail128 ? n_pass - 1 : n_pass;
    // option pair256 (scan_topk256.hip, ABL bit 128): the main passes two at a time in ONE launch — each pass on half the grid, the
    // partner workgroups of the two passes on one XCD walking the same tiles, so that a corpus line leaves HBM once per 512
    // queries.  A pass of a paired launch has grid / 2 lists per query, stored behind each other: such passes are merged in
    // groups of their own; an odd main pass is left over and runs unpaired.
    // (bench-only: the ablation ladder of the paired headline instantiation, ablate 1 / 5 / 7 / 9 at d = 768 and lists of 64)
    const bool pair_ablate_ok = opt.ablate == 0 || ((opt.ablate == 1 || opt.ablate == 5 || opt.ablate == 7 || opt.ablate == 9) && dp == 768 && kp == 64);
    const bool pair256 = grouped && use256 && opt.pair256 != 0 && pair_ablate_ok && grid == 256 && n_main >= 2 &&
                         bh_scan256_pair_supports(dp, kp) && (dp != 1024 || opt.ring_variant == 0 || opt.ring_variant == 5);
    const int n_paired = pair256 ? (n_main & ~1) : 0;
    struct Group {
        int p0, p1;
        bool paired;
    };
    std::vector<Group> groups;
    const int n_paired_full = bal_first >= 0 ? bal_first : n_paired;  // (the balanced pair is merged pass by pass: its passes do not start bq apart)
    for (int g0 = 0; g0 < n_paired_full; g0 += 2 * group) groups.push_back({g0, std::min(n_paired_full, g0 + 2 * group), true});
    if (bal_first >= 0) groups.push_back({bal_first, bal_first + 2, true});
    for (int g0 = n_paired; g0 < n_main; g0 += group) groups.push_back({g0, std::min(n_main, g0 + group), false});
    const int n_group = grouped ? (int)groups.size() + (tail128 ? 1 : 0) : n_pass;
    for (int p = 0; p < n_group; ++p)
        if (!ix->event(2 + 4 * p + 3)) return fail(BH_EHIP, "hipEventCreate failed");

    HIP_TRY(hipEventRecord(ev_begin, st));
    // every pass's slot tables and bounds start at ordf(-inf); its claim counter (scan_topk256 dynamic tile distribution)
    // starts at the first tile that is not handed out by workgroup index
    HIP_TRY(bh_launch_fill_u32(ix->gthr.p, (long long)(gthr_pass * (size_t)n_pass), 0x007fffffu, st, use256 ? (long long)gthr_pass : 0,
                               (long long)bq * (BH_SLOTS256 + 1), use256 ? bh_scan256_first_claimed_tile((int)ix->n_tiles, grid, dp) : 0u));
    if (n_paired > 0)  // (a pass of a paired launch is distributed over half the grid)
        HIP_TRY(bh_launch_fill_u32(ix->gthr.p, (long long)(gthr_pass * (size_t)n_paired), 0x007fffffu, st, (long long)gthr_pass,
                                   (long long)bq * (BH_SLOTS256 + 1), bh_scan256_first_claimed_tile((int)ix->n_tiles, grid / 2, dp)));
    double alg_bytes = 0;
    auto scan_args = [&](int p, bh_u64* partial_p) {
        const int q0 = passes[p].first, qs = passes[p].second;
        BhScanArgs sa;
        sa.corpus = ix->rows;
        sa.n_rows = ix->n_rows;
        sa.n_tiles = ix->n_tiles;
        sa.qtile = ix->qbuf.p + (size_t)q0 * dp;
        sa.cand = ix->cand.p;
        sa.partial = partial_p;
        sa.gthr = ix->gthr.p + gthr_pass * (size_t)p;
        sa.share = opt.share_threshold;
        sa.nontemporal = opt.nontemporal;
        sa.ablate = opt.ablate;
        sa.ring_variant = opt.ring_variant;
        sa.qsplit = qs;
        sa.dma_interleave = opt.dma_interleave;
        sa.pair_window = opt.pair_window;
        sa.dyn_tiles = opt.dyn_tiles;
        sa.clk = use256 ? ix->clk.p : nullptr;
        sa.progress = ix->gthr.p + progress_off;  // [grid] words behind the threshold blocks
        return sa;
    };
    auto merge_args = [&](int q0, int tile, int n_lists, const bh_u64* partial_p, long long pass_stride) {
        BhMergeArgs ma;
        ma.partial = partial_p;
        ma.pass_stride = pass_stride;
        ma.n_lists = n_lists;
        ma.bq = tile;
        ma.corpus = ix->rows;
        ma.n_rows = ix->n_rows;
        ma.qtile = ix->qbuf.p + (size_t)q0 * dp;
        ma.dim_padded = dp;
        ma.k = k;
        ma.id_offset = id_offset;
        ma.out_scores = out_scores_dev + (size_t)q0 * k;
        ma.out_ids = reinterpret_cast<long long*>(out_ids_dev) + (size_t)q0 * k;
        ma.err_coef = err_coef;
        ma.uncert = opt.certify ? ix->uncert.p + q0 : nullptr;
        ma.kth_key = opt.certify ? ix->kth.p + q0 : nullptr;
        ma.n_uncert = opt.certify ? ix->n_uncert_dev : nullptr;
        return ma;
    };
    auto launch_scan = [&](const BhScanArgs& sa) {
        if (use256) return bh_launch_scan256(sa, dp, kp, grid, st);
        if (use192) return bh_launch_scan192(sa, dp, kp, grid, st);
        return bh_launch_scan(sa, dp, kp, qw, grid, st);
    };
    if (grouped) {
        for (int gi = 0; gi < (int)groups.size(); ++gi) {
            const int g0 = groups[(size_t)gi].p0, g1 = groups[(size_t)gi].p1;
            const bool paired = groups[(size_t)gi].paired;
            const size_t pass_elems = paired ? partial_elems / 2 : partial_elems;  // keys between the list sets of consecutive passes
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * gi), st));
            for (int p = g0; p < g1; p += paired ? 2 : 1) {
                BhScanArgs sa = scan_args(p, ix->partial.p + (size_t)(p - g0) * pass_elems);
                if (paired) {  // passes p and p + 1: queries, threshold blocks and list sets behind each other
                    if (p == bal_first) {
                        sa.nq_valid[0] = bal_nq[0];
                        sa.nq_valid[1] = bal_nq[1];
                        sa.qtile2 = ix->qbuf.p + (size_t)passes[(size_t)p + 1].first * dp;
                    } else if (opt.balance_tail != 0) {
                        sa.nq_valid[0] = bq;
                        sa.nq_valid[1] = std::min(bq, nq - passes[(size_t)p + 1].first);
                    }  // (balance_tail = 0: {0, 0} = every wave scans, the kernel behaviour before the balanced remainder)
                    sa.qsplit = 2;
                    sa.pair_window = opt.pair256 == 2 ? 0 : 1;
                    // the partner finds a line in L2 only if the first reader's request left it there: the stream of a paired
                    // launch is cached unless pair256 = 3 (same box: 84.9 ms per headline search non-temporal, 83.0 cached; unpaired 88.4 / 90.3)
                    if (opt.pair256 != 3) sa.nontemporal = 0;
                    HIP_TRY(hipMemsetAsync(sa.progress, 0, (size_t)grid * sizeof(unsigned), st));
                    HIP_TRY(bh_launch_scan256_paired(sa, dp, kp, grid, st));
                } else {
                    if (opt.balance_tail != 0) sa.nq_valid[0] = std::min(bq, nq - passes[(size_t)p].first);
                    HIP_TRY(launch_scan(sa));
                }
                // per pass  N*d*2 (corpus, read once) + Bq*d*2 + Bq*k*12, with the LOGICAL d
                alg_bytes += (paired ? 2.0 : 1.0) * ((double)ix->n_rows * ix->dim * 2.0 + (double)bq * ix->dim * 2.0 + (double)bq * k * 12.0);
            }
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * gi + 1), st));
            const int q0 = passes[g0].first;
            const int nq_group = std::min(nq, passes[g1 - 1].first + bq) - q0;
            if (g0 == bal_first) {  // the balanced pair: one merge per pass (the second pass's queries start bal_nq[0], not bq, behind the first's)
                HIP_TRY(bh_launch_merge_rescore(merge_args(q0, bq, grid / 2, ix->partial.p, 0), kp, bal_nq[0], st));
                HIP_TRY(bh_launch_merge_rescore(merge_args(passes[(size_t)g0 + 1].first, bq, grid / 2, ix->partial.p + pass_elems, 0), kp, bal_nq[1], st));
            } else
            HIP_TRY(bh_launch_merge_rescore(merge_args(q0, bq, paired ? grid / 2 : grid, ix->partial.p, (long long)pass_elems), kp, nq_group, st));
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * gi + 3), st));
        }
        if (tail128) {
            const int gi = n_group - 1, p = n_pass - 1, q0 = passes[p].first;
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * gi), st));
            // The tail pass runs a DIFFERENT kernel (scan_topk.hip, 128 queries) inside buffers sized for the 256-query kernel:
            // its needs are stated and checked here instead of being implied by the other kernel's layout.
            //   thresholds  [128][64] words at ordf(-inf): the head of this pass's threshold block, which the fill above set
            //               — the block's claim counter (the one word that is not ordf(-inf)) sits behind the slot tables
            //   candidates  [grid][128][2 * kp] keys, result lists [grid][128][kp] keys
            const size_t tail_gthr = (size_t)128 * 64, tail_cand = (size_t)grid * 128 * 2 * kp, tail_lists = (size_t)grid * 128 * kp;
            if (tail_gthr > (size_t)bq * (BH_SLOTS256 + 1) || gthr_pass * (size_t)p + tail_gthr > ix->gthr.cap ||
                tail_cand > ix->cand.cap || tail_lists > ix->partial.cap)
                return fail(BH_EHIP, "internal: the 128-query tail pass does not fit the 256-query kernel's buffers");
            BhScanArgs sa{};
            sa.corpus = ix->rows;
            sa.n_rows = ix->n_rows;
            sa.n_tiles = ix->n_tiles;
            sa.qtile = ix->qbuf.p + (size_t)q0 * dp;
            sa.cand = ix->cand.p;
            sa.partial = ix->partial.p;  // (stream order: the main groups' merges are done with the lists)
            sa.gthr = ix->gthr.p + gthr_pass * (size_t)p;
            sa.share = opt.share_threshold;
            sa.nontemporal = opt.nontemporal;
            sa.ablate = 0;
            sa.ring_variant = 0;
            sa.qsplit = 1;
            sa.progress = nullptr;  // (paired workgroups only)
            sa.dma_interleave = opt.dma_interleave;
            sa.pair_window = 0;
            sa.dyn_tiles = 0;
            sa.clk = nullptr;
            HIP_TRY(bh_launch_scan(sa, dp, kp, 1, grid, st));
            alg_bytes += (double)ix->n_rows * ix->dim * 2.0 + 128.0 * ix->dim * 2.0 + 128.0 * k * 12.0;
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * gi + 1), st));
            HIP_TRY(bh_launch_merge_rescore(merge_args(q0, 128, grid, ix->partial.p, 0), kp, nq - q0, st));
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * gi + 3), st));
        }
    } else {
        for (int p = 0; p < n_pass; ++p) {
            const int q0 = passes[p].first, qs = passes[p].second;
            const int tile = bq * qs;
            const int nq_tile = std::min(tile, nq - q0);
            bh_u64* partial_p = ix->partial.p + (size_t)(p & 1) * partial_elems;
            const BhScanArgs sa = scan_args(p, partial_p);
            if (qs > 1) HIP_TRY(hipMemsetAsync(sa.progress, 0, (size_t)grid * sizeof(unsigned), st));
            // this pass overwrites the partial set that pass p - 2 left for its merge: wait for that merge
            if (p >= 2) HIP_TRY(hipStreamWaitEvent(st, ix->event(2 + 4 * (p - 2) + 3), 0));
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * p), st));
            HIP_TRY(launch_scan(sa));
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * p + 1), st));
            // merge + canonical re-score on the side stream, beside the next pass's scan
            HIP_TRY(hipStreamWaitEvent(ms, ix->event(2 + 4 * p + 1), 0));
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * p + 2), ms));
            HIP_TRY(bh_launch_merge_rescore(merge_args(q0, tile, grid / qs, partial_p, 0), kp, nq_tile, ms));
            HIP_TRY(hipEventRecord(ix->event(2 + 4 * p + 3), ms));
            alg_bytes += (double)ix->n_rows * ix->dim * 2.0 + (double)tile * ix->dim * 2.0 + (double)tile * k * 12.0;
        }
        // the search ends when the last merges have: bring the side stream back into the main one
        for (int p = std::max(0, n_pass - 2); p < n_pass; ++p) HIP_TRY(hipStreamWaitEvent(st, ix->event(2 + 4 * p + 3), 0));
    }
    HIP_TRY(hipEventRecord(ev_end, st));
    HIP_TRY(spin_sync(st, (ix->last_shape_nq == nq && ix->last_shape_k == k && ix->last_shape_rows == ix->n_rows) ? ix->last_shape_ms : 0.0));
    // ---- exactness: queries the certificate could not prove go through the exact scan (certify.hip)
    int64_t n_uncert = 0, n_filter_passes = 0, n_filter_rows = 0;
    double exact_ms = 0;
    if (opt.certify && *ix->n_uncert_host != 0u) {
        std::vector<unsigned> flags((size_t)nq);
        HIP_TRY(hipMemcpyAsync(flags.data(), ix->uncert.p, (size_t)nq * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<int> todo;
        for (int q = 0; q < nq; ++q)
            if (flags[(size_t)q]) todo.push_back(q);
        n_uncert = (int64_t)todo.size();
        if (!todo.empty()) {
            hipEvent_t x0 = ix->event(2 + 4 * (size_t)n_group), x1 = ix->event(3 + 4 * (size_t)n_group);
            if (!x0 || !x1) return fail(BH_EHIP, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(x0, st));
            // one FILTER PASS per batch of uncertified queries: the scan's stream + MFMA loop with the fixed threshold
            // k-th canonical score - error bound  lists every row that can still belong to the top k — 256 queries per pass on
            // scan_topk256.hip (ABL bit 64) where it applies (option filter256; d = 384 / 512 / 768), else 128 on scan_topk.hip
            // (ABL 5) —, bh_exact_rescore_kernel gives the listed rows their canonical scores, the host sorts
            const bool f256 = opt.filter256 != 0 && bh_filter256_supports(dp) && grid % 8 == 0;
            const int ftile = f256 ? 256 : 128;
            if ((rc = ix->exact_q.ensure((size_t)BH_EXACT_BATCH * dp))) return rc;
            if ((rc = ix->exact_keys.ensure((size_t)BH_EXACT_BATCH * BH_EXACT_CAP + BH_EXACT_BATCH))) return rc;
            if ((rc = ix->exact_rows.ensure((size_t)BH_EXACT_BATCH * BH_EXACT_CAP))) return rc;
            if ((rc = ix->exact_cnt.ensure(4 * BH_EXACT_BATCH))) return rc;  // counts | thresholds (float) | query indices | spare
            bh_u64* kth_dev = ix->exact_keys.p + (size_t)BH_EXACT_BATCH * BH_EXACT_CAP;
            unsigned* cnt_dev = ix->exact_cnt.p;
            float* thr_dev = reinterpret_cast<float*>(ix->exact_cnt.p + BH_EXACT_BATCH);
            int* todo_dev = reinterpret_cast<int*>(ix->exact_cnt.p + 2 * BH_EXACT_BATCH);
            std::vector<bh_u64> keys;
            std::vector<float> st_s(todo.size() * (size_t)k);       // the re-done lists, in `todo` order
            std::vector<long long> st_i(todo.size() * (size_t)k);
            for (size_t b0 = 0; b0 < todo.size(); b0 += (size_t)ftile) {
                const int nb = (int)std::min<size_t>((size_t)ftile, todo.size() - b0);
                HIP_TRY(hipMemcpyAsync(todo_dev, todo.data() + b0, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemsetAsync(cnt_dev, 0, BH_EXACT_BATCH * sizeof(unsigned), st));
                HIP_TRY(bh_launch_exact_prepare(ix->qbuf.p, todo_dev, nb, ftile, ix->kth.p, err_coef, dp, ix->exact_q.p, kth_dev, thr_dev, st));
                BhScanArgs fa = scan_args(0, nullptr);
                fa.qtile = ix->exact_q.p;
                fa.cand = nullptr;
                fa.gthr = nullptr;
                fa.share = 0;
                fa.ablate = 0;
                fa.ring_variant = 0;
                fa.qsplit = 1;
                fa.dma_interleave = 1;
                fa.clk = nullptr;
                fa.fix_thr = thr_dev;
                fa.fix_cnt = cnt_dev;
                fa.fix_rows = ix->exact_rows.p;
                fa.fix_cap = BH_EXACT_CAP;
                fa.dyn_tiles = 0;
                fa.pair_window = 0;
                fa.progress = nullptr;
                if (f256 && use256 && bq == 256 && opt.dyn_tiles != 0) {
                    // the claimed tail of the corpus (scan_topk256's dynamic tile distribution): the filter pass borrows the first
                    // threshold block of the search — its passes are over (stream order) — for the claim counter
                    HIP_TRY(bh_launch_fill_u32(ix->gthr.p, (long long)gthr_pass, 0x007fffffu, st, (long long)gthr_pass, (long long)bq * (BH_SLOTS256 + 1),
                                               bh_scan256_first_claimed_tile((int)ix->n_tiles, grid, dp)));
                    fa.gthr = ix->gthr.p;
                    fa.dyn_tiles = 1;
                }
                if (f256)
                    HIP_TRY(bh_launch_filter_scan256(fa, dp, grid, st));
                else
                    HIP_TRY(bh_launch_filter_scan(fa, dp, grid, st));
                BhExactArgs ea;
                ea.corpus = ix->rows;
                ea.n_rows = ix->n_rows;
                ea.dim_padded = dp;
                ea.q = ix->exact_q.p;
                ea.nqf = nb;
                ea.kth_key = kth_dev;
                ea.rows = ix->exact_rows.p;
                ea.cnt = cnt_dev;
                ea.out_keys = ix->exact_keys.p;
                HIP_TRY(bh_launch_exact_rescore(ea, st));
                unsigned cnt[BH_EXACT_BATCH];
                HIP_TRY(hipMemcpyAsync(cnt, cnt_dev, sizeof cnt, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                unsigned max_cnt = 0;
                for (int j = 0; j < nb; ++j) {
                    if (cnt[j] > BH_EXACT_CAP)
                        return fail(BH_EUNSUPPORTED, "query %d: more than %d rows reach its k-th score within rounding error "
                                    "(%u): exact fall-back list overflow", todo[b0 + j], BH_EXACT_CAP, cnt[j]);
                    max_cnt = std::max(max_cnt, cnt[j]);
                }
                // the batch's key lists in ONE strided copy (a copy per query cost ~10 us each: 3 of them per uncertified query
                // were a third of the fall-back's time on the clustered corpus)
                keys.resize((size_t)nb * max_cnt);
                if (max_cnt > 0)
                    HIP_TRY(hipMemcpy2D(keys.data(), (size_t)max_cnt * sizeof(bh_u64), ix->exact_keys.p, (size_t)BH_EXACT_CAP * sizeof(bh_u64),
                                        (size_t)max_cnt * sizeof(bh_u64), (size_t)nb, hipMemcpyDeviceToHost));
                for (int j = 0; j < nb; ++j) {
                    n_filter_rows += cnt[j];
                    bh_u64* kq = keys.data() + (size_t)j * max_cnt;
                    const size_t take = std::min<size_t>((size_t)k, cnt[j]);
                    // canonical order: score desc, row asc (0 = did not qualify: last)
                    std::partial_sort(kq, kq + take, kq + cnt[j], std::greater<bh_u64>());
                    float* rs = st_s.data() + (b0 + (size_t)j) * k;
                    long long* ri = st_i.data() + (b0 + (size_t)j) * k;
                    for (int t = 0; t < k; ++t) {
                        if ((size_t)t < take && kq[(size_t)t] != 0ull) {
                            const unsigned o = (unsigned)(kq[(size_t)t] >> 32);
                            const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
                            memcpy(&rs[t], &u, 4);
                            ri[t] = id_offset + (long long)(unsigned)~(unsigned)kq[(size_t)t];
                        } else {
                            rs[t] = -INFINITY;
                            ri[t] = -1;
                        }
                    }
                }
                ++n_filter_passes;
            }
            // the re-done lists go to their rows of the result buffers (device memory or pinned host memory alike) in one launch
            {
                const size_t nt = todo.size();
                const size_t bytes_s = nt * k * sizeof(float), bytes_i = nt * k * sizeof(long long), off_i = (bytes_s + 15) & ~(size_t)15,
                             off_t = off_i + ((bytes_i + 15) & ~(size_t)15);
                if ((rc = ix->staging.ensure(off_t + nt * sizeof(int)))) return rc;
                HIP_TRY(hipMemcpyAsync(ix->staging.p, st_s.data(), bytes_s, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(ix->staging.p + off_i, st_i.data(), bytes_i, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(ix->staging.p + off_t, todo.data(), nt * sizeof(int), hipMemcpyHostToDevice, st));
                HIP_TRY(bh_launch_scatter_lists(reinterpret_cast<const float*>(ix->staging.p), reinterpret_cast<const long long*>(ix->staging.p + off_i),
                                                reinterpret_cast<const int*>(ix->staging.p + off_t), (int)nt, k, out_scores_dev,
                                                reinterpret_cast<long long*>(out_ids_dev), st));
            }
            HIP_TRY(hipEventRecord(x1, st));
            HIP_TRY(hipStreamSynchronize(st));
            float xm = 0;
            HIP_TRY(hipEventElapsedTime(&xm, x0, x1));
            exact_ms = xm;
        }
    }

    bh_counters& c = ix->counters;
    c.n_rows = ix->n_rows;
    c.dim = ix->dim;
    c.dim_padded = dp;
    c.query_tile = (tail128 && n_pass == 1) ? 128 : bq * qs_max;  // (a search of at most 128 queries ran on the 128-query kernel alone)
    c.tail_scan_ms = 0;
    c.tail_query_tile = 0;
    if (tail128) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ix->event(2 + 4 * (n_group - 1)), ix->event(2 + 4 * (n_group - 1) + 1)));
        c.tail_scan_ms = ms;
        c.tail_query_tile = 128;
    }
    c.n_passes = n_pass;
    c.n_workgroups = grid;
    c.k_padded = kp;
    c.scan_ms = 0;
    c.merge_ms = 0;
    c.paired_scan_ms = 0;
    c.paired_launches = n_paired / 2;
    c.balanced_scan_ms = 0;
    c.balanced_queries = bal_first >= 0 ? bal_nq[0] + bal_nq[1] : 0;
    for (int p = 0; p < n_group; ++p) {
        float ms = 0;
        // grouped: the group's scans back to back (launch gaps included), then its one merge; paired: per pass, the merge
        // on the side stream (beside the next pass's scan: not additive with scan_ms)
        HIP_TRY(hipEventElapsedTime(&ms, ix->event(2 + 4 * p), ix->event(2 + 4 * p + 1)));
        c.scan_ms += ms;
        if (grouped && p < (int)groups.size() && groups[(size_t)p].paired) c.paired_scan_ms += ms;
        if (grouped && p < (int)groups.size() && bal_first >= 0 && groups[(size_t)p].p0 == bal_first) c.balanced_scan_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ix->event(2 + 4 * p + (grouped ? 1 : 2)), ix->event(2 + 4 * p + 3)));
        c.merge_ms += ms;
    }
    float tot = 0;
    HIP_TRY(hipEventElapsedTime(&tot, ev_begin, ev_end));
    c.total_ms = tot + exact_ms;
    ix->last_shape_nq = nq;
    ix->last_shape_k = k;
    ix->last_shape_rows = ix->n_rows;
    ix->last_shape_ms = tot;
    c.algorithmic_bytes = alg_bytes;
    c.uncertified_queries = n_uncert;
    c.exact_ms = exact_ms;
    c.exact_passes = n_filter_passes;
    c.exact_rows_rescored = n_filter_rows;
    c.shader_mhz = 0;
    if (use256) {  // effective shader clock of the last scan launch: cycles per 100 MHz tick, averaged over the workgroups
        std::vector<bh_u64> h((size_t)grid * 8);
        HIP_TRY(hipMemcpy(h.data(), ix->clk.p, h.size() * sizeof(bh_u64), hipMemcpyDeviceToHost));
        double cyc = 0, ticks = 0;
        for (int g = 0; g < grid; ++g) {  // (tile loop: words 4 -> 5 in cycles, 1 -> 2 in ticks)
            cyc += (double)(h[8 * g + 5] - h[8 * g + 4]);
            ticks += (double)(h[8 * g + 2] - h[8 * g + 1]);
        }
        if (ticks > 0) c.shader_mhz = 100.0 * cyc / ticks;
    }
    return BH_OK;
}

namespace {

// k > 248 (the reference accepts any top_k_documents, modules/retrieve.py:157): the candidate lists of the scan kernels hold
// at most 256 entries, so the corpus is cut into contiguous row RANGES small enough that no range is expected to hold more
// than ~100 of the top k, every range is searched for its exact top 248 by the ordinary fused search (a view of the index:
// same kernels, same certificate), and the ranges' lists are merged in canonical order.  This is exact iff no range holds
// more than 248 of the true top k, which is CHECKED: a range whose list is full (248 entries) and whose last entry still
// belongs to the merged top k may have dropped a row — it is split in four and searched again, down to ranges of one
// 32-row tile (which cannot overflow a 248-entry list).  Bytes streamed: one corpus pass per tile of 128 queries
// (candidate lists of 256), as for k = 248; the price is the per-range fixed cost and a host merge of S x 248 entries per query.
int search_large_k(bh_index* ix, const void* q_dev, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
                   float* out_scores_dev, int64_t* out_ids_dev) {
    if (!q_dev || !out_scores_dev || !out_ids_dev) return fail(BH_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(ix->device));
    constexpr int KK = BH_MAX_LIST_K;
    // A range holds lists for a SUBSET of the queries once it has been split: only the queries whose list overflowed in the
    // parent are searched again in the four children (round 3 searched them for all nq queries: on clustered data with many
    // queries that multiplied corpus passes and host time); the parent keeps serving the others.
    struct Range {
        int64_t t0, t1;           // tiles [t0, t1)
        std::vector<int> qs;      // the queries this range holds lists for (ascending)
        std::vector<int> qpos;    // [nq] position of query q in qs, -1 = none
        std::vector<float> s;     // [qs.size()][KK]
        std::vector<long long> i;
    };
    _Float16* const rows0 = ix->rows;
    const int64_t n_rows0 = ix->n_rows, n_tiles0 = ix->n_tiles;
    const size_t qrow_bytes = (size_t)ix->dim * (q_dtype == BH_F16 ? 2 : 4);
    float* d_s = nullptr;
    long long* d_i = nullptr;
    unsigned char* d_qsub = nullptr;  // gathered query rows of a subset search
    HIP_TRY(hipMalloc((void**)&d_s, (size_t)nq * KK * sizeof(float)));
    if (hipMalloc((void**)&d_i, (size_t)nq * KK * sizeof(long long)) != hipSuccess ||
        hipMalloc((void**)&d_qsub, (size_t)nq * qrow_bytes) != hipSuccess) {
        (void)hipFree(d_s);
        if (d_i) (void)hipFree(d_i);
        return fail(BH_ENOMEM, "hipMalloc of the range lists");
    }
    bh_counters total{};
    int rc = BH_OK;
    auto search_range = [&](Range& r) -> int {
        const int nqs = (int)r.qs.size();
        const void* qptr = q_dev;
        if (nqs != nq) {  // gather the subset's query rows (few: the queries one range overflowed for)
            for (int j = 0; j < nqs; ++j)
                HIP_TRY(hipMemcpyAsync(d_qsub + (size_t)j * qrow_bytes, (const unsigned char*)q_dev + (size_t)r.qs[(size_t)j] * qrow_bytes, qrow_bytes,
                                       hipMemcpyDeviceToDevice, ix->stream));
            HIP_TRY(hipStreamSynchronize(ix->stream));
            qptr = d_qsub;
        }
        const int64_t row_lo = r.t0 * 32, row_hi = std::min<int64_t>(n_rows0, r.t1 * 32);
        ix->rows = rows0 + (size_t)row_lo * ix->dim_padded;  // a view: rows of the range, tile-aligned
        ix->n_rows = row_hi - row_lo;
        ix->n_tiles = r.t1 - r.t0;
        const int rc2 = search_device_lists(ix, qptr, q_dtype, nqs, KK, id_offset + row_lo, d_s, reinterpret_cast<int64_t*>(d_i));
        ix->rows = rows0;
        ix->n_rows = n_rows0;
        ix->n_tiles = n_tiles0;
        if (rc2 != BH_OK) return rc2;
        const bh_counters& c = ix->counters;
        total.scan_ms += c.scan_ms;
        total.merge_ms += c.merge_ms;
        total.total_ms += c.total_ms;
        total.algorithmic_bytes += c.algorithmic_bytes;
        total.n_passes += c.n_passes;
        total.uncertified_queries += c.uncertified_queries;
        total.exact_ms += c.exact_ms;
        total.exact_passes += c.exact_passes;
        total.exact_rows_rescored += c.exact_rows_rescored;
        total.paired_scan_ms += c.paired_scan_ms;
        total.paired_launches += c.paired_launches;
        total.balanced_scan_ms += c.balanced_scan_ms;
        total.balanced_queries += c.balanced_queries;
        total.query_tile = c.query_tile;
        total.n_workgroups = c.n_workgroups;
        total.k_padded = c.k_padded;
        total.shader_mhz = c.shader_mhz;
        r.s.resize((size_t)nqs * KK);
        r.i.resize((size_t)nqs * KK);
        HIP_TRY(hipMemcpy(r.s.data(), d_s, r.s.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(r.i.data(), d_i, r.i.size() * sizeof(long long), hipMemcpyDeviceToHost));
        return BH_OK;
    };
    auto make_range = [&](int64_t t0, int64_t t1, const std::vector<int>& qs) {
        Range r{t0, t1, qs, std::vector<int>((size_t)nq, -1), {}, {}};
        for (size_t j = 0; j < qs.size(); ++j) r.qpos[(size_t)qs[j]] = (int)j;
        return r;
    };
    // ranges: ~100 expected members of the top k each, tile-aligned
    std::vector<Range> ranges;
    {
        std::vector<int> all_q((size_t)nq);
        for (int q = 0; q < nq; ++q) all_q[(size_t)q] = q;
        const int64_t want = std::max<int64_t>(1, std::min<int64_t>(n_tiles0, (k + 99) / 100));
        const int64_t per = (n_tiles0 + want - 1) / want;
        for (int64_t t = 0; t < n_tiles0; t += per) ranges.push_back(make_range(t, std::min(n_tiles0, t + per), all_q));
    }
    for (auto& r : ranges)
        if ((rc = search_range(r)) != BH_OK) break;
    std::vector<float> out_s((size_t)nq * k);
    std::vector<long long> out_i((size_t)nq * k);
    struct Ent {
        float s;
        long long id;
    };
    std::vector<Ent> all;
    constexpr int kMaxRounds = 64;
    for (int round = 0; rc == BH_OK; ++round) {
        std::vector<std::vector<int>> over(ranges.size());  // per range: the queries whose list may have dropped a member of the top k
        bool any = false;
        for (int q = 0; q < nq; ++q) {
            all.clear();
            for (size_t j = 0; j < ranges.size(); ++j) {
                const int at = ranges[j].qpos[(size_t)q];
                if (at < 0) continue;
                for (int t = 0; t < KK; ++t) {
                    const long long id = ranges[j].i[(size_t)at * KK + t];
                    if (id >= 0) all.push_back(Ent{ranges[j].s[(size_t)at * KK + t], id});
                }
            }
            std::sort(all.begin(), all.end(), [](const Ent& a, const Ent& b) { return a.s != b.s ? a.s > b.s : a.id < b.id; });
            const size_t take = std::min<size_t>(all.size(), (size_t)k);
            for (size_t t = 0; t < (size_t)k; ++t) {
                out_s[(size_t)q * k + t] = t < take ? all[t].s : -INFINITY;
                out_i[(size_t)q * k + t] = t < take ? all[t].id : -1;
            }
            // a FULL list whose last entry is still inside the merged top k may have dropped a member of the top k
            for (size_t j = 0; j < ranges.size(); ++j) {
                const int at = ranges[j].qpos[(size_t)q];
                if (at < 0) continue;
                const long long last_id = ranges[j].i[(size_t)at * KK + KK - 1];
                if (last_id < 0) continue;  // the range's list is not full: it holds every candidate of the range
                const float last_s = ranges[j].s[(size_t)at * KK + KK - 1];
                // (fewer than k entries in all: every full list may hide members of the top k)
                const bool last_in_topk = take < (size_t)k || last_s > all[take - 1].s ||
                                          (last_s == all[take - 1].s && last_id <= all[take - 1].id);
                if (last_in_topk && ranges[j].t1 - ranges[j].t0 > 1) {
                    over[j].push_back(q);
                    any = true;
                }
            }
        }
        if (!any) break;
        if (round + 1 >= kMaxRounds) {  // (cannot happen: a range shrinks fourfold per round down to one tile, which cannot overflow)
            rc = fail(BH_EUNSUPPORTED, "k = %d: the range search did not converge in %d rounds", k, kMaxRounds);
            break;
        }
        std::vector<Range> next;
        for (size_t j = 0; j < ranges.size() && rc == BH_OK; ++j) {
            if (over[j].empty()) {
                next.push_back(std::move(ranges[j]));
                continue;
            }
            const int64_t span = ranges[j].t1 - ranges[j].t0, per = (span + 3) / 4;
            for (int64_t t = ranges[j].t0; t < ranges[j].t1 && rc == BH_OK; t += per) {
                Range r = make_range(t, std::min(ranges[j].t1, t + per), over[j]);
                rc = search_range(r);
                next.push_back(std::move(r));
            }
            // the parent keeps its lists for the queries it did not overflow for
            for (int q : over[j]) ranges[j].qpos[(size_t)q] = -1;
            if (over[j].size() < ranges[j].qs.size()) next.push_back(std::move(ranges[j]));
        }
        ranges.swap(next);
    }
    (void)hipFree(d_qsub);
    (void)hipFree(d_s);
    (void)hipFree(d_i);
    if (rc != BH_OK) return rc;
    HIP_TRY(hipMemcpy(out_scores_dev, out_s.data(), out_s.size() * sizeof(float), hipMemcpyDefault));
    HIP_TRY(hipMemcpy(out_ids_dev, out_i.data(), out_i.size() * sizeof(long long), hipMemcpyDefault));
    total.n_rows = n_rows0;
    total.dim = ix->dim;
    total.dim_padded = ix->dim_padded;
    ix->counters = total;
    return BH_OK;
}

}  // namespace

int bh_search(bh_index* ix, const void* q_host, int32_t q_dtype, int32_t nq, int32_t k, int64_t id_offset,
              float* out_scores, int64_t* out_ids) {
    if (!ix) return fail(BH_EINVAL, "null index");
    if (nq < 0 || k <= 0) return fail(BH_EINVAL, "nq=%d k=%d", nq, k);
    if (q_dtype != BH_F16 && q_dtype != BH_F32) return fail(BH_EINVAL, "bad q_dtype %d", q_dtype);
    if (nq == 0) return bh_search_device(ix, nullptr, q_dtype, 0, k, id_offset, nullptr, nullptr);
    if (!q_host || !out_scores || !out_ids) return fail(BH_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(ix->device));
    const size_t esz = q_dtype == BH_F16 ? 2 : 4;
    const size_t qbytes = (size_t)nq * ix->dim * esz;
    const size_t sbytes = (size_t)nq * k * sizeof(float), ibytes = (size_t)nq * k * sizeof(int64_t);
    unsigned char* tmp = nullptr;
    const size_t off_s = (qbytes + 255) & ~(size_t)255, off_i = off_s + ((sbytes + 255) & ~(size_t)255);
    HIP_TRY(hipMalloc((void**)&tmp, off_i + ibytes));
    int rc = BH_OK;
    hipError_t e = hipMemcpy(tmp, q_host, qbytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(BH_EHIP, "H2D queries: %s", hipGetErrorString(e));
    if (rc == BH_OK)
        rc = bh_search_device(ix, tmp, q_dtype, nq, k, id_offset, (float*)(tmp + off_s), (int64_t*)(tmp + off_i));
    if (rc == BH_OK) {
        e = hipMemcpy(out_scores, tmp + off_s, sbytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out_ids, tmp + off_i, ibytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(BH_EHIP, "D2H results: %s", hipGetErrorString(e));
    }
    (void)hipFree(tmp);
    return rc;
}

int bh_merge_topk_device(const float* scores_dev, const int64_t* ids_dev, int32_t n_lists, int32_t nq, int32_t k,
                         float* out_scores_dev, int64_t* out_ids_dev) {
    if (n_lists <= 0 || nq < 0 || k <= 0) return fail(BH_EINVAL, "n_lists=%d nq=%d k=%d", n_lists, nq, k);
    if (nq == 0) return BH_OK;
    if (!scores_dev || !ids_dev || !out_scores_dev || !out_ids_dev) return fail(BH_EINVAL, "null buffer");
    if (k > 4096) return fail(BH_EUNSUPPORTED, "k=%d too large", k);
    const int group = std::max(1, 4096 / k);  // lists merged per launch
    if (n_lists <= group) {
        HIP_TRY(bh_launch_merge_lists(scores_dev, reinterpret_cast<const long long*>(ids_dev), n_lists, nq, k,
                                      out_scores_dev, reinterpret_cast<long long*>(out_ids_dev), nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        return BH_OK;
    }
    if (group < 2) {
        // k > 2048: two lists no longer fit one launch of the merge kernel (4096 keys).  The lists are short and few (one per
        // shard / rank): merge them on the host — a k-way merge of sorted lists in the canonical order, ids < 0 skipped —
        // exactly like search_large_k merges its ranges.
        const size_t per = (size_t)nq * k;
        std::vector<float> hs((size_t)n_lists * per), os(per);
        std::vector<long long> hi((size_t)n_lists * per), oi(per);
        HIP_TRY(hipMemcpy(hs.data(), scores_dev, hs.size() * sizeof(float), hipMemcpyDefault));
        HIP_TRY(hipMemcpy(hi.data(), ids_dev, hi.size() * sizeof(long long), hipMemcpyDefault));
        std::vector<std::pair<float, long long>> all;
        for (int q = 0; q < nq; ++q) {
            all.clear();
            for (int l = 0; l < n_lists; ++l)
                for (int t = 0; t < k; ++t) {
                    const size_t at = (size_t)l * per + (size_t)q * k + t;
                    if (hi[at] >= 0) all.emplace_back(hs[at], hi[at]);
                }
            const size_t take = std::min<size_t>(all.size(), (size_t)k);
            std::partial_sort(all.begin(), all.begin() + take, all.end(),
                              [](const std::pair<float, long long>& a, const std::pair<float, long long>& b) {
                                  return a.first != b.first ? a.first > b.first : a.second < b.second;
                              });
            for (size_t t = 0; t < (size_t)k; ++t) {
                os[(size_t)q * k + t] = t < take ? all[t].first : -INFINITY;
                oi[(size_t)q * k + t] = t < take ? all[t].second : -1;
            }
        }
        HIP_TRY(hipMemcpy(out_scores_dev, os.data(), os.size() * sizeof(float), hipMemcpyDefault));
        HIP_TRY(hipMemcpy(out_ids_dev, oi.data(), oi.size() * sizeof(long long), hipMemcpyDefault));
        return BH_OK;
    }
    // tree reduction: merge `group` lists at a time into a scratch set of lists
    const int n_mid = (n_lists + group - 1) / group;
    float* mid_s = nullptr;
    long long* mid_i = nullptr;
    HIP_TRY(hipMalloc((void**)&mid_s, (size_t)n_mid * nq * k * sizeof(float)));
    hipError_t e = hipMalloc((void**)&mid_i, (size_t)n_mid * nq * k * sizeof(long long));
    if (e != hipSuccess) {
        (void)hipFree(mid_s);
        return fail(BH_ENOMEM, "hipMalloc: %s", hipGetErrorString(e));
    }
    int rc = BH_OK;
    for (int g = 0; g < n_mid && rc == BH_OK; ++g) {
        const int l0 = g * group, nl = std::min(group, n_lists - l0);
        e = bh_launch_merge_lists(scores_dev + (size_t)l0 * nq * k,
                                  reinterpret_cast<const long long*>(ids_dev) + (size_t)l0 * nq * k, nl, nq, k,
                                  mid_s + (size_t)g * nq * k, mid_i + (size_t)g * nq * k, nullptr);
        if (e != hipSuccess) rc = fail(BH_EHIP, "merge launch: %s", hipGetErrorString(e));
    }
    if (rc == BH_OK) {
        e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) rc = fail(BH_EHIP, "merge: %s", hipGetErrorString(e));
    }
    if (rc == BH_OK)
        rc = bh_merge_topk_device(mid_s, reinterpret_cast<const int64_t*>(mid_i), n_mid, nq, k, out_scores_dev,
                                  out_ids_dev);
    (void)hipFree(mid_s);
    (void)hipFree(mid_i);
    return rc;
}

int bh_merge_topk(const float* scores, const int64_t* ids, int32_t n_lists, int32_t nq, int32_t k, float* out_scores,
                  int64_t* out_ids) {
    if (n_lists <= 0 || nq < 0 || k <= 0) return fail(BH_EINVAL, "n_lists=%d nq=%d k=%d", n_lists, nq, k);
    if (nq == 0) return BH_OK;
    if (!scores || !ids || !out_scores || !out_ids) return fail(BH_EINVAL, "null buffer");
    const size_t n_in = (size_t)n_lists * nq * k, n_out = (size_t)nq * k;
    float* d_s = nullptr;
    HIP_TRY(hipMalloc((void**)&d_s, (n_in + n_out) * (sizeof(float) + sizeof(int64_t)) + 8));
    int64_t* d_i = reinterpret_cast<int64_t*>(d_s + n_in + n_out + ((n_in + n_out) & 1));
    // layout: [scores in | scores out | pad] [ids in | ids out]
    float* d_so = d_s + n_in;
    int64_t* d_io = d_i + n_in;
    int rc = BH_OK;
    hipError_t e = hipMemcpy(d_s, scores, n_in * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_i, ids, n_in * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(BH_EHIP, "H2D: %s", hipGetErrorString(e));
    if (rc == BH_OK) rc = bh_merge_topk_device(d_s, d_i, n_lists, nq, k, d_so, d_io);
    if (rc == BH_OK) {
        e = hipMemcpy(out_scores, d_so, n_out * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out_ids, d_io, n_out * sizeof(int64_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(BH_EHIP, "D2H: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_s);
    return rc;
}

int64_t bh_debug_scan_timeline(const bh_index* ix, uint64_t* out, int64_t max_words) {
    if (!ix || !out || max_words < 0) return fail(BH_EINVAL, "null argument");
    const int64_t n = std::min<int64_t>(max_words, (int64_t)ix->clk.cap);
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipMemcpy(out, ix->clk.p, (size_t)n * sizeof(bh_u64), hipMemcpyDeviceToHost));
    return n;
}

int bh_bench_counters(const bh_index* ix, bh_counters* out) {
    if (!ix || !out) return fail(BH_EINVAL, "null argument");
    if (!bh_copy_sized(out, ix->counters, 16))
        return fail(BH_EINVAL, "bh_counters.struct_size = %d: set it to sizeof(bh_counters) before the call (BH_VERSION %d)", out->struct_size, BH_VERSION);
    return BH_OK;
}

}  // extern "C"
