// C ABI of libgrpg_rasterizer.so (include/grpg_rasterizer.h): host orchestration of the HIP
// pipeline.  Mirrors CudaRasterizer::Rasterizer::{forward,backward,markVisible,visible_filter}
// (cuda_rasterizer/rasterizer_impl.cu:197-343, :396-505, :141-153, :345-392).
//
// Forward schedule on the caller's stream:
//   frame init -> preprocess (+ per-workgroup sums of the tile counts) -> count publish (num_rendered
//   and the coarse pair count land in a pinned host word, event) -> depth sort of the (key,id)
//   pairs -> binning (hierarchical: coarse scan / emit / partition / counts / fill; or sort: offsets
//   scan / instance emit / stable tile partition / ranges) -> render (-> semantic render).
// The binning blob is carved for a capacity remembered from earlier frames, every kernel reads the
// instance count from device memory, and the host reads the published count only AFTER the whole
// frame is enqueued (it needs it for the return value): by then the count -- known 70 us into the
// frame -- has arrived, so neither the stream nor the host idles.  First frame of a shape / exact
// mode: read the count first, then carve and enqueue the tail; capacity overflow: redo the tail,
// like the reference's one blocking read (rasterizer_impl.cu:284).
// There is no CPU fallback: without a usable HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <atomic>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/grpg_rasterizer.h"
#include "common.h"

using namespace grpg;

namespace {

thread_local std::string g_last_error;
thread_local int g_timing_enabled = 0;   // 0 off, 1 all stages, 2 render stage only

// ---- num_rendered hand-over: pinned, device-mapped host words + one event per (thread, device) ----
struct HostWord {
  int dev = -1;
  uint32_t* host_ptr = nullptr;   // [0] num_rendered, [1] coarse pairs, [3] sticky asynchronous error
  uint32_t* dev_ptr = nullptr;    // the same memory as the device addresses it
  hipEvent_t ev = nullptr;
};
// released when the owning thread exits (as the deferred-frame ring below)
// ... except on the MAIN thread: its thread_local destructors run during process exit, when the HIP
// runtime may already be shutting down (Python finalisation with the torch extension loaded); calling
// into it then is a known source of exit-time errors, and the OS reclaims the memory anyway.
static bool on_main_thread() { return (long)syscall(SYS_gettid) == (long)getpid(); }
struct HostWords : std::vector<HostWord> {
  ~HostWords() {
    if (on_main_thread()) return;
    for (auto& h : *this) {
      if (h.host_ptr) (void)hipHostFree(h.host_ptr);
      if (h.ev) (void)hipEventDestroy(h.ev);
    }
  }
};
thread_local HostWords g_host_words;

HostWord* host_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (auto& h : g_host_words)
    if (h.dev == dev) return &h;
  HostWord h;
  h.dev = dev;
  if (hipHostMalloc((void**)&h.host_ptr, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
  if (hipHostGetDevicePointer((void**)&h.dev_ptr, h.host_ptr, 0) != hipSuccess ||
      hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipHostFree(h.host_ptr);
    return nullptr;
  }
  h.host_ptr[0] = 0u; h.host_ptr[1] = 0u; h.host_ptr[2] = 0u; h.host_ptr[3] = 0u;
  g_host_words.push_back(h);
  return &g_host_words.back();
}

// ---- capacity hints: high-water mark of num_rendered per (device, P, W, H), process-wide ----
struct CapKey { int dev, P, W, H; };

// ---- deferred frames (grpg_forward_deferred / grpg_frame_status): the count of a frame lands in
// its own pinned words; nobody waits for it until its status is asked for.  A ring per thread;
// a slot that comes round again while still pending is resolved first. ----
struct DeferSlot {
  int dev = -1;
  uint32_t* host_ptr = nullptr;
  uint32_t* dev_ptr = nullptr;
  hipEvent_t ev = nullptr;
  int state = 0;            // 0 free, 1 pending (count not looked at yet), 2 resolved
  int result = 0;           // resolved: num_rendered or GRPG_ERR_CAPACITY
  uint32_t Rcap = 0, Ccap = 0;
  bool hier = false;
  bool with_pass3 = true;   // the frame was enqueued with the fourth depth-sort pass
  CapKey key{-1, 0, 0, 0};
  int generation = 0;       // ticket = generation * DEFER_SLOTS + slot: a recycled slot's old tickets expire
};
constexpr int DEFER_SLOTS = GRPG_MAX_DEFERRED_FRAMES;
// The ring belongs to the thread that enqueues the frames: tickets are only valid on that thread
// (documented in the header), and the pinned words / events are released when the thread exits.
struct DeferRing {
  DeferSlot slot[DEFER_SLOTS];
  ~DeferRing() {
    if (on_main_thread()) return;   // process exit: see HostWords
    for (auto& d : slot) {
      if (d.host_ptr) (void)hipHostFree(d.host_ptr);
      if (d.ev) (void)hipEventDestroy(d.ev);
      d.host_ptr = nullptr; d.ev = nullptr;
    }
  }
  DeferSlot& operator[](int i) { return slot[i]; }
};
thread_local DeferRing g_defer;
thread_local int g_defer_next = 0;
struct CapHint {
  CapKey key{-1, 0, 0, 0};
  uint32_t high = 0;      // decaying high-water mark of num_rendered
  uint32_t high_c = 0;    // same for the coarse (Gaussian, super-tile) count of the hierarchical binning
  bool far = true;        // a recent frame's depth keys needed the fourth sort pass (sort.hip key_far)
  uint32_t far_hold = 0;  // near frames left before `far` is dropped again (sticky with decay, below)
  uint32_t once = 0, once_c = 0;   // grpg_set_capacity_hint: exact capacities for the next frame only
  bool seen = false;      // a frame of this shape has reported its counts (high / high_c / far are real)
  uint64_t stamp = 0;     // last use (LRU replacement)
  bool valid = false;
};
std::mutex g_hint_mu;
CapHint g_hints[16];
uint64_t g_hint_clock = 0;
std::atomic<int> g_binning_mode{[] {
  const char* e = getenv("GRPG_SYNC_R");
  return (e && atoi(e) != 0) ? GRPG_BINNING_EXACT : GRPG_BINNING_SPECULATIVE;
}()};

bool same_key(const CapKey& a, const CapKey& b) {
  return a.dev == b.dev && a.P == b.P && a.W == b.W && a.H == b.H;
}
// capacity carved for an expected count: +25 % head room, 64 Ki granules (stable blob sizes, so the
// caller's caching allocator hands the same block back every frame)
uint32_t padded_capacity(uint32_t r) {
  const uint64_t c = ((uint64_t)r + r / 4 + 65536 + 65535) / 65536 * 65536;
  return (uint32_t)(c > 0x7FFF0000ull ? 0x7FFF0000ull : c);
}
uint32_t capacity_from_hint(const CapKey& k, uint32_t* coarse_cap, bool* far) {
  *far = true;   // no history: enqueue the (normally idle) fourth depth-sort pass
  std::lock_guard<std::mutex> lk(g_hint_mu);
  for (auto& h : g_hints)
    if (h.valid && same_key(h.key, k)) {
      h.stamp = ++g_hint_clock;
      if (h.once != 0u) {   // grpg_set_capacity_hint: exactly what the caller promised, once
        const uint32_t cap = h.once, cc = h.once_c;
        h.once = 0u; h.once_c = 0u;
        *coarse_cap = cc;
        *far = true;
        return cap;
      }
      if (!h.seen) break;   // only ever seeded: no history yet
      const uint32_t cap = padded_capacity(h.high);
      const uint32_t cc = h.high_c ? padded_capacity(h.high_c) : cap;
      *coarse_cap = cc < cap ? cc : cap;     // a (Gaussian, super-tile) pair holds >= 1 instance
      *far = h.far;
      return cap;
    }
  *coarse_cap = 0u;
  return 0u;
}
void update_hint(const CapKey& k, uint32_t R, uint32_t Rc, bool far) {
  std::lock_guard<std::mutex> lk(g_hint_mu);
  CapHint* slot = nullptr;
  for (auto& h : g_hints)
    if (h.valid && same_key(h.key, k)) { slot = &h; break; }
  if (!slot) {
    slot = &g_hints[0];
    for (auto& h : g_hints) {
      if (!h.valid) { slot = &h; break; }
      if (h.stamp < slot->stamp) slot = &h;
    }
    *slot = CapHint{};
    slot->key = k;
    slot->valid = true;
  }
  const uint32_t decayed = slot->high - slot->high / 64;   // lets the mark follow a shrinking scene
  slot->high = R > decayed ? R : decayed;
  // decays on sort-binning frames too (Rc == 0 there), so it never outlives `high`
  const uint32_t dc = slot->high_c - slot->high_c / 64;
  slot->high_c = Rc > dc ? Rc : dc;
  if (slot->high_c > slot->high) slot->high_c = slot->high;
  // Sticky with decay: frames alternating between a near and a far depth range under one
  // (device, P, W, H) -- two cameras, two scenes of one size -- would otherwise enqueue every far frame
  // without its fourth pass and render it twice.  An enqueued-but-idle fourth pass costs two launch
  // boundaries (~9 us of a single stream); it is dropped after FAR_HOLD consecutive near frames.
  constexpr uint32_t FAR_HOLD = 32;
  if (far) slot->far_hold = FAR_HOLD;
  else if (slot->far_hold > 0u) slot->far_hold--;
  slot->far = slot->far_hold > 0u;
  slot->seen = true;
  slot->stamp = ++g_hint_clock;
}

// The fat depth sort (sort.hip) sweeps the whole count table in every workgroup: fine up to a few
// hundred chunks, quadratic beyond (P > 4 M: the classic three-kernel passes).
bool depth_sort_is_fat(uint32_t nchunks_ds) { return nchunks_ds <= DS_MAX_CHUNKS; }

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(GRPG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

// debug=true: synchronise and check after every stage, like CHECK_CUDA (auxiliary.h:166-173).
#define STAGE_CHECK(name)                                                                    \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize(stream);                        \
    if (e_ != hipSuccess)                                                                    \
      return fail(GRPG_ERR_HIP, std::string("stage ") + name + ": " + hipGetErrorString(e_)); \
  } while (0)

// Asynchronous device-side failures (today: a producer / consumer hand-over of the render that timed
// out, render_fwd.hip pc_fail) raise a sticky word in pinned host memory.  Like an asynchronous HIP
// error it is reported by the NEXT entry point the thread calls (all forwards, the backwards,
// grpg_frame_status), once; with debug = true the forward that suffered it fails itself.
int check_async_error(HostWord* hw) {
  if (hw && hw->host_ptr[3] != 0u) {
    hw->host_ptr[3] = 0u;
    return fail(GRPG_ERR_HIP, "render: a producer/consumer hand-over timed out in an earlier frame "
                              "of this thread (image blob header pc_timeout): that frame's image is invalid");
  }
  return GRPG_OK;
}

int ensure_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(GRPG_ERR_NO_DEVICE,
                "no usable HIP device (libgrpg_rasterizer has no CPU fallback by design)");
  return GRPG_OK;
}

// Tile binning algorithm: "hier" (hier_binning.hip, default) or "sort" (emit + stable partition).
std::atomic<int> g_binning_alg{[] {
  const char* e = getenv("GRPG_BINNING");
  return (e && strcmp(e, "sort") == 0) ? GRPG_BINNING_ALG_SORT : GRPG_BINNING_ALG_HIER;
}()};
bool binning_is_hier() { return g_binning_alg.load() == GRPG_BINNING_ALG_HIER; }
// -DGRPG_UNFUSED_COARSE_EMIT (experiment build `unfusedemit`): the coarse emit behind the two-launch offsets scan, as
// before round 5 (still the path of grids beyond 255 x 255 tiles), for A/B runs on one box
#ifdef GRPG_UNFUSED_COARSE_EMIT
constexpr bool UNFUSED_COARSE_EMIT = true;
#else
constexpr bool UNFUSED_COARSE_EMIT = false;
#endif

int bits_for(uint32_t T) {  // smallest b with (1 << b) >= T, i.e. tile ids fit in b bits
  int b = 0;
  while ((1ull << b) < (unsigned long long)T) b++;
  return b;
}

// Per-stage device timing with HIP events recorded on the op's stream.  Nothing is synchronised
// while frames are in flight: every grpg_forward appends one record of events to a thread-local
// list; grpg_get_stage_timing() waits for them, sums per stage and clears the list.
struct TimingRecord {
  hipEvent_t ev[GRPG_NUM_STAGES + 1];
  int stage_of[GRPG_NUM_STAGES + 1];
  int n = 0;
};
thread_local std::vector<TimingRecord*> g_pending;
thread_local std::vector<TimingRecord*> g_free;

struct StageTimer {
  hipStream_t s;
  TimingRecord* r = nullptr;
  int mode = 0;
  StageTimer(hipStream_t s_, int mode_) : s(s_), mode(mode_) {
    if (!mode) return;
    if (!g_free.empty()) { r = g_free.back(); g_free.pop_back(); }
    else { r = new TimingRecord(); for (auto& e : r->ev) (void)hipEventCreate(&e); }
    r->n = 0;
  }
  int tail_n = -1;   // record count when the tail of the frame (everything redone on overflow) began
  void begin_tail() { if (r && tail_n < 0) tail_n = r->n; }
  void restart_tail() { if (r && tail_n >= 0) r->n = tail_n; }
  void mark(int next_stage) {
    if (!r || r->n > GRPG_NUM_STAGES) return;
    if (mode == 2 && next_stage != 6 && next_stage != 7) return;   // render start / render end only
    (void)hipEventRecord(r->ev[r->n], s);
    r->stage_of[r->n] = next_stage;
    r->n++;
  }
  void finish() {
    if (r) { g_pending.push_back(r); r = nullptr; }
  }
  ~StageTimer() {
    if (r) g_free.push_back(r);   // call failed before finish(): recycle
  }
};

// Backward: two device times per call (blend backward, preprocess backward), same mechanism -- but
// process-wide: PyTorch runs an op's backward on its autograd thread, not on the thread that
// switched the timing on.
struct BwdTimingRecord { hipEvent_t ev[3]; };
std::atomic<int> g_bwd_timing{0};
std::mutex g_bwd_mu;
std::vector<BwdTimingRecord*> g_bwd_pending;
std::vector<BwdTimingRecord*> g_bwd_free;
BwdTimingRecord* bwd_timing_begin(hipStream_t s) {
  if (!g_bwd_timing.load()) return nullptr;
  BwdTimingRecord* r = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_bwd_mu);
    if (!g_bwd_free.empty()) { r = g_bwd_free.back(); g_bwd_free.pop_back(); }
  }
  if (!r) { r = new BwdTimingRecord(); for (auto& e : r->ev) (void)hipEventCreate(&e); }
  (void)hipEventRecord(r->ev[0], s);
  return r;
}
constexpr size_t BWD_PENDING_CAP = 256;   // a caller that never collects them keeps the newest 256
void bwd_timing_mark(BwdTimingRecord* r, int i, hipStream_t s) {
  if (!r) return;
  (void)hipEventRecord(r->ev[i], s);
  if (i == 2) {
    std::lock_guard<std::mutex> lk(g_bwd_mu);
    if (g_bwd_pending.size() >= BWD_PENDING_CAP) {   // recycle the oldest record (its events are reused)
      g_bwd_free.push_back(g_bwd_pending.front());
      g_bwd_pending.erase(g_bwd_pending.begin());
    }
    g_bwd_pending.push_back(r);
  }
}
// a backward that fails between begin and the last mark hands its record back
struct BwdTimingGuard {
  BwdTimingRecord* r;
  bool done = false;
  explicit BwdTimingGuard(BwdTimingRecord* r_) : r(r_) {}
  ~BwdTimingGuard() {
    if (r && !done) { std::lock_guard<std::mutex> lk(g_bwd_mu); g_bwd_free.push_back(r); }
  }
};

// ---- which geometry blobs were carved WITH the backward's gradient records (a training forward)?
// grpg_backward writes P x 64 bytes behind the rest of the blob; a blob carved by an evaluation
// forward (GRPG_FORWARD_NO_BACKWARD) has no such region.  The last forwards' blob addresses are
// remembered so that the mistake is an error return, not an out-of-bounds device write. ----
struct GeomSeen { const void* ptr = nullptr; int P = 0; bool has_grad = false; uint64_t stamp = 0; };
std::mutex g_geom_mu;
GeomSeen g_geom_seen[64];
uint64_t g_geom_clock = 0;
void remember_geom(const void* ptr, int P, bool has_grad) {
  std::lock_guard<std::mutex> lk(g_geom_mu);
  GeomSeen* slot = &g_geom_seen[0];
  for (auto& g : g_geom_seen) {
    if (g.ptr == ptr) { slot = &g; break; }
    if (g.stamp < slot->stamp) slot = &g;
  }
  slot->ptr = ptr; slot->P = P; slot->has_grad = has_grad; slot->stamp = ++g_geom_clock;
}
// The table is ADVISORY: every forward of this process overwrites the entry of the address it carves
// (an evaluation forward at a recycled address turns a "yes" into a "no"), but a blob the CALLER copies into an
// address a training forward of the same P once carved is invisible to it -- the one case left in which the fast
// path vouches for a buffer it has not seen (validating every call would cost the training loop a host wait per
// backward: the host could no longer run ahead of the device).  Only its "yes, the training forward of this P
// carved that address" answer is taken as is (the training loop's case: no host round trip).  A "no" (possibly stale: a reloaded training blob
// at an address an evaluation forward used) or an unknown address is settled by the geometry header
// itself -- one 4-byte read behind the stream's work, off the hot path.
// returns GRPG_OK, or the error the backward must return
int geom_check_backward(const void* ptr, int P, hipStream_t stream) {
  {
    std::lock_guard<std::mutex> lk(g_geom_mu);
    for (auto& g : g_geom_seen)
      if (g.ptr == ptr && g.P == P && g.has_grad) return GRPG_OK;
  }
  // An address the table does not vouch for: the blob's own header decides.  That is a host read behind the
  // stream's work -- grpg_backward is NOT asynchronous for such a blob (documented in grpg_rasterizer.h), and a
  // stream that is being captured into a graph cannot be waited for at all: refuse instead of breaking the capture.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(GRPG_ERR_BAD_BUFFER, "grpg_backward under stream capture needs the geometry buffer of a training "
                                     "forward made by this process (its address is not on record, and reading the "
                                     "header would synchronize the capturing stream)");
  BlobHeader h;
  hipError_t e = hipMemcpyAsync(&h, ptr, sizeof(BlobHeader), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return fail(GRPG_ERR_HIP, std::string("reading the geometry header: ") + hipGetErrorString(e));
  if (h.magic != GEOM_MAGIC || h.P != (uint32_t)P)
    return fail(GRPG_ERR_BAD_BUFFER, "the geometry buffer does not hold a forward of this P (magic / P)");
  if (h.has_grad_rec == 0u)
    return fail(GRPG_ERR_BAD_BUFFER, "the geometry buffer was carved by an evaluation forward "
                                     "(GRPG_FORWARD_NO_BACKWARD): it has no room for the backward's gradient records");
  remember_geom(ptr, P, true);
  return GRPG_OK;
}

CameraArgs make_camera(const float* view, const float* proj, const float* campos, int W, int H,
                       float tan_fovx, float tan_fovy) {
  CameraArgs c;
  c.view = view; c.proj = proj; c.campos = campos;
  c.W = W; c.H = H;
  c.gx = (W + TILE - 1) / TILE;
  c.gy = (H + TILE - 1) / TILE;
  c.tan_fovx = tan_fovx; c.tan_fovy = tan_fovy;
  c.focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:225-226
  c.focal_x = W / (2.0f * tan_fovx);
  return c;
}

}  // namespace

extern "C" {

int grpg_abi_version(void) { return GRPG_ABI_VERSION; }
const char* grpg_last_error(void) { return g_last_error.c_str(); }

int grpg_set_stage_timing(int enabled) {
  g_timing_enabled = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
  for (auto* r : g_pending) g_free.push_back(r);
  g_pending.clear();
  g_bwd_timing.store(enabled != 0 ? 1 : 0);
  {
    std::lock_guard<std::mutex> lk(g_bwd_mu);
    for (auto* r : g_bwd_pending) g_bwd_free.push_back(r);
    g_bwd_pending.clear();
  }
  return GRPG_OK;
}
int grpg_get_stage_timing(float* stage_ms_sum, int* num_calls) {
  if (!stage_ms_sum || !num_calls) return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL output");
  for (int i = 0; i < GRPG_NUM_STAGES; i++) stage_ms_sum[i] = 0.f;
  *num_calls = 0;
  for (auto* r : g_pending) {
    if (r->n < 2) { g_free.push_back(r); continue; }
    hipError_t e = hipEventSynchronize(r->ev[r->n - 1]);
    if (e != hipSuccess) return fail(GRPG_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    for (int i = 0; i + 1 < r->n; i++) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, r->ev[i], r->ev[i + 1]);
      if (r->stage_of[i] >= 0 && r->stage_of[i] < GRPG_NUM_STAGES) stage_ms_sum[r->stage_of[i]] += ms;
    }
    (*num_calls)++;
    g_free.push_back(r);
  }
  g_pending.clear();
  return GRPG_OK;
}

int grpg_get_backward_timing(float* blend_ms_sum, float* preprocess_ms_sum, int* num_calls) {
  if (!blend_ms_sum || !preprocess_ms_sum || !num_calls) return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL output");
  *blend_ms_sum = 0.f; *preprocess_ms_sum = 0.f; *num_calls = 0;
  std::lock_guard<std::mutex> lk(g_bwd_mu);
  for (auto* r : g_bwd_pending) {
    hipError_t e = hipEventSynchronize(r->ev[2]);
    if (e != hipSuccess) return fail(GRPG_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, r->ev[0], r->ev[1]);
    (void)hipEventElapsedTime(&b, r->ev[1], r->ev[2]);
    *blend_ms_sum += a; *preprocess_ms_sum += b; (*num_calls)++;
    g_bwd_free.push_back(r);
  }
  g_bwd_pending.clear();
  return GRPG_OK;
}

int grpg_set_binning_mode(int mode) {
  if (mode != GRPG_BINNING_SPECULATIVE && mode != GRPG_BINNING_EXACT)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "unknown binning mode");
  g_binning_mode.store(mode);
  return GRPG_OK;
}
int grpg_get_binning_algorithm(void) { return g_binning_alg.load(); }
int grpg_set_binning_algorithm(int alg) {
  if (alg != GRPG_BINNING_ALG_SORT && alg != GRPG_BINNING_ALG_HIER)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "unknown binning algorithm");
  g_binning_alg.store(alg);
  return GRPG_OK;
}
int grpg_reset_capacity_hints(void) {
  std::lock_guard<std::mutex> lk(g_hint_mu);
  for (auto& h : g_hints) h = CapHint{};
  return GRPG_OK;
}
int grpg_set_capacity_hint(int P, int width, int height, unsigned instances, unsigned coarse_pairs) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P <= 0 || width <= 0 || height <= 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "bad shape");
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const CapKey k = {dev, P, width, height};
  std::lock_guard<std::mutex> lk(g_hint_mu);
  CapHint* slot = nullptr;
  for (auto& h : g_hints)
    if (h.valid && same_key(h.key, k)) { slot = &h; break; }
  if (!slot) {
    slot = &g_hints[0];
    for (auto& h : g_hints) {
      if (!h.valid) { slot = &h; break; }
      if (h.stamp < slot->stamp) slot = &h;
    }
    *slot = CapHint{};
    slot->key = k;
    slot->valid = true;
  }
  slot->once = instances ? instances : 1u;
  slot->once_c = coarse_pairs ? coarse_pairs : 1u;
  slot->stamp = ++g_hint_clock;
  return GRPG_OK;
}

}  // extern "C"

namespace {

// Pinned staging for the composed forward's segment table (thread-local; the frame's one host wait
// guarantees the previous upload has been consumed before the buffer is reused).
thread_local SegmentDev* g_seg_staging = nullptr;

// grpg_backward_composed returns without a host wait, so the pinned tables its asynchronous copies
// read must outlive the call: a ring of slots, each guarded by an event recorded behind its copies
// (a slot comes round again 8 backward calls later; its event has long fired by then).
struct BwdStagingSlot {
  SegmentDev* segs = nullptr;
  grpg_model_segment_grad* grads = nullptr;
  hipEvent_t ev = nullptr;
  bool used = false;
};
constexpr int BWD_STAGING_SLOTS = 8;
thread_local BwdStagingSlot g_bwd_staging[BWD_STAGING_SLOTS];
thread_local int g_bwd_staging_next = 0;
BwdStagingSlot* bwd_staging_acquire() {
  BwdStagingSlot& b = g_bwd_staging[g_bwd_staging_next];
  g_bwd_staging_next = (g_bwd_staging_next + 1) % BWD_STAGING_SLOTS;
  if (!b.segs) {
    if (hipHostMalloc((void**)&b.segs, sizeof(SegmentDev) * MAX_SEGMENTS, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&b.grads, sizeof(grpg_model_segment_grad) * MAX_SEGMENTS, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess)
      return nullptr;
  }
  if (b.used) (void)hipEventSynchronize(b.ev);
  b.used = true;
  return &b;
}

// grpg_forward_layers: the class of every Gaussian and the two extra plane pairs
struct LayerArgs {
  const unsigned char* segment_class;   // composed frames: [num_segments] HOST bytes (NULL: the segment's rigid flag)
  const unsigned char* layer_class;
  const float* layer_background;
  float* out_color_bg; float* out_alpha_bg; float* out_color_obj; float* out_alpha_obj;
};

// Shared body of grpg_forward (segs == NULL: flat input tensors) and grpg_forward_composed (segs:
// per-model raw parameters, P = sum of their counts, the flat pointers are NULL).
int forward_impl(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                 void* binning_user, grpg_alloc_fn image_alloc, void* image_user, int P, int D,
                 int M, int S, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* semantics, const float* opacities, const float* scales,
                 float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                 float tan_fovx, float tan_fovy, float* out_color,
                 float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug,
                 void* hip_stream, const grpg_model_segment* segs, int nseg, unsigned flags = 0u,
                 DeferSlot* defer = nullptr, bool force_pass3 = false, const LayerArgs* layers = nullptr,
                 const FrameEpilogue* epi = nullptr) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P < 0 || width <= 0 || height <= 0 || S < 0 || M < 0)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "negative size");
  if (!geometry_alloc || !binning_alloc || !image_alloc)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "buffer allocators must not be NULL");
  if (!background || !viewmatrix || !projmatrix || !cam_pos)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL camera/background/output pointer");
  // the float planes may be left out only when a frame epilogue delivers the bytes instead (grpg_forward_frame)
  if ((!out_color || !out_depth || !out_alpha) && !(epi && !epi->planes && epi->rgb8))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL camera/background/output pointer");
  if (epi && ((flags & GRPG_FORWARD_NO_BACKWARD) == 0u || S != 0))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "a frame epilogue needs an evaluation frame without semantic planes");
  if ((unsigned)P > ID_MASK) return fail(GRPG_ERR_INVALID_ARGUMENT, "P must be < 2^28");
  if (P > 0 && segs == nullptr) {
    if (!means3D || !opacities) return fail(GRPG_ERR_INVALID_ARGUMENT, "means3D/opacities NULL");
    if (!cov3D_precomp && (!scales || !rotations))
      return fail(GRPG_ERR_INVALID_ARGUMENT, "need scales+rotations or cov3D_precomp");
    if (!colors_precomp && (!shs || M <= 0))
      return fail(GRPG_ERR_INVALID_ARGUMENT, "need shs (M>0) or colors_precomp");
    if (!colors_precomp && (D < 0 || D > 3 || (D + 1) * (D + 1) > M))
      return fail(GRPG_ERR_INVALID_ARGUMENT, "SH degree needs (D+1)^2 <= M, D <= 3");
    if (S > 0 && (!semantics || !out_semantic))
      return fail(GRPG_ERR_INVALID_ARGUMENT, "semantics/out_semantic NULL with S>0");
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  const CameraArgs cam = make_camera(viewmatrix, projmatrix, cam_pos, width, height, tan_fovx, tan_fovy);
  const uint32_t T = (uint32_t)cam.gx * (uint32_t)cam.gy;
  const size_t N = (size_t)width * height;

  // an evaluation frame (no backward can follow) carves the blob without the gradient records
  const GeomLayout GL = geom_layout((size_t)P, (flags & GRPG_FORWARD_NO_BACKWARD) == 0u);
  const ImgLayout IL = img_layout(T, N);
  char* geom = geometry_alloc(GL.total, geometry_user);
  char* img = image_alloc(IL.total, image_user);
  if (!geom || !img) return fail(GRPG_ERR_ALLOC, "geometry/image buffer allocation failed");
  remember_geom(geom, P, (flags & GRPG_FORWARD_NO_BACKWARD) == 0u);

  float4* rec_w = (float4*)(geom + GL.rec);
  const RecView rec = {rec_w};
  uint32_t* key_a = (uint32_t*)(geom + GL.key_a);
  uint32_t* key_b = (uint32_t*)(geom + GL.key_b);
  uint32_t* val_a = (uint32_t*)(geom + GL.val_a);
  uint32_t* val_b = (uint32_t*)(geom + GL.val_b);
  uint32_t* tiles = (uint32_t*)(geom + GL.tiles);
  uint32_t* offsets = (uint32_t*)(geom + GL.offsets);
  int* radii_int = radii ? radii : (int*)(geom + GL.radii);
  uint32_t* table = (uint32_t*)(geom + GL.table);
  uint32_t* totals = (uint32_t*)(geom + GL.totals);
  uint32_t* block_sums = (uint32_t*)(geom + GL.block_sums);
  uint32_t* ds_table = (uint32_t*)(geom + GL.ds_table);
  BlobHeader* gh = (BlobHeader*)geom;
  uint2* ranges = (uint2*)(img + IL.ranges);
  uint32_t* n_contrib = (uint32_t*)(img + IL.n_contrib);
  uint32_t* work = (uint32_t*)(img + IL.work);

  // A frame whose bytes go to pinned host memory is DRAINED (render_fwd.hip FrameEpi) when its width allows whole
  // 16-byte pieces per tile row (W % 16 == 0; whole 64-byte lines per unit when W % 64 == 0): staging bytes in the n_contrib plane (an evaluation frame tracks no n_contrib: 4 bytes
  // per pixel lie idle), the units' arrival counters behind the layered frame's tile flags in the checkpoint-count
  // area (a training forward's; T / 4 of its 4 T words).  GRPG_DRAIN_WGS (experiments): workgroups that carry the
  // units, 0 = the blending waves store straight into host memory.
  FrameEpilogue epi_drained;
  uint32_t drain_cnt_words = 0u;
  if (epi && epi->rgb8 && epi->rgb8_host && (width & 15) == 0 && P > 0) {
    static const int drain_cfg = [] { const char* v = getenv("GRPG_DRAIN_WGS"); return v ? atoi(v) : FRAME_DRAIN_WGS; }();
    // every unit needs a slot (64 per drain wave, RW_WAVES waves per workgroup), the counters 3 T words at most
    const uint32_t NU = (uint32_t)((width + 63) / 64) * (uint32_t)cam.gy;
    uint32_t wgs = (uint32_t)drain_cfg > (NU + 255u) / 256u ? (uint32_t)drain_cfg : (NU + 255u) / 256u;
    if (wgs > 3u * T / 256u) wgs = 3u * T / 256u;
    if (drain_cfg > 0 && wgs > 0u && wgs * 256u >= NU) {
      epi_drained = *epi;
      epi_drained.drain_stage = (unsigned char*)(img + IL.n_contrib);
      epi_drained.drain_cnt = (uint32_t*)(img + IL.ck_count) + T;
      epi_drained.drain_wgs = (int)wgs;
      drain_cnt_words = wgs * 256u;
      epi = &epi_drained;
    }
  }

  StageTimer tm(stream, g_timing_enabled);
  uint32_t R = 0;

  if (P > 0) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HostWord* hw = host_word();
    if (!hw) return fail(GRPG_ERR_HIP, "pinned host word / event allocation failed");
    if (int rc = check_async_error(hw)) return rc;
    // where the speculative frame publishes its count: the thread's word, or the deferred frame's own
    uint32_t* const pub_ptr = defer ? defer->dev_ptr : hw->dev_ptr;
    hipEvent_t const pub_ev = defer ? defer->ev : hw->ev;
    // Instance capacity of the binning blob.  Speculative mode (default): from the high-water mark
    // of earlier frames of this (device, P, W, H), so the blob is carved and every kernel of the
    // frame is enqueued BEFORE the host learns num_rendered; the kernels read the count from device
    // memory.  Exact mode, or no history yet: wait for the count first, like the reference
    // (rasterizer_impl.cu:284).
    const bool fat_sort = depth_sort_is_fat(GL.nchunks_ds);
    // hierarchical binning needs the compacted depth order of the fat sort (its coarse scan gathers
    // the super-tile counts by sorted id); otherwise emit + partition
    const uint32_t NS = super_tiles(cam.gx, cam.gy);
    const bool hier = fat_sort && binning_is_hier() && NS <= 65536u;   // 16 id bits in a coarse key
    // cap: instances (point list); ccap: coarse (Gaussian, super-tile) pairs, <= cap
    auto layout_for = [&](uint32_t cap, uint32_t ccap) {
      return hier ? bin_layout((size_t)cap, T, NS, ccap) : bin_layout((size_t)cap);
    };
    // a training forward leaves blend checkpoints of its long tile lists behind the binning blob
    // (common.h CK_*; the backward starts its segments from them)
    const bool with_ckpt = (flags & GRPG_FORWARD_NO_BACKWARD) == 0u && S == 0;
    // work-list classes of the frame's render.  A layered frame: the plain frame's classes, plus one flag per tile
    // -- does its list hold an object entry (preprocess sets the flags; they live in the image blob's checkpoint-count
    // area, which only a training forward uses) -- that sends short tiles WITH objects down the quarter-wave path
    unsigned char* const tile_obj = layers ? (unsigned char*)(img + IL.ck_count) : nullptr;
    const uint32_t tile_obj_words = layers ? (T + 3u) / 4u : 0u;
    const TileClasses frame_classes = layers ? tile_classes_layers(tile_obj, cam.gx) : tile_classes(S);
    const LayerMarks marks = {layers ? layers->layer_class : nullptr, tile_obj, cam.gx};
    auto blob_bytes = [&](const BinLayout& L, uint32_t cap) {
      return L.total + (with_ckpt ? ckpt_bytes(ckpt_slots(cap)) : 0);
    };
    const CapKey ck = {dev, P, width, height};
    uint32_t Ccap = 0u;
    bool hint_far = true;
    uint32_t Rcap = g_binning_mode.load() == GRPG_BINNING_SPECULATIVE ? capacity_from_hint(ck, &Ccap, &hint_far) : 0u;
    const bool speculative = Rcap != 0u;
    // Fourth pass of the depth sort (sort.hip): needed when the visible depth keys span >= 2^27
    // values -- a property of the scene's scale that does not change from frame to frame.  Its two
    // launches are only enqueued when the previous frame of this shape needed them (or nothing is
    // known yet); a frame that turns out to need them without having them is rendered again.
    const bool with_pass3 = force_pass3 || !speculative || hint_far;
    char* bin = nullptr;
    BinLayout BL{};
    if (speculative) {
      BL = layout_for(Rcap, Ccap);
      bin = binning_alloc(blob_bytes(BL, Rcap), binning_user);
      if (!bin) return fail(GRPG_ERR_ALLOC, "binning buffer allocation failed");
    }
    tm.mark(0);
    launch_frame_init(stream, geom, bin, img, (uint32_t)P, fat_sort ? 0u : (uint32_t)P, Rcap,
                      (uint32_t)width, (uint32_t)height, (uint32_t)S, ranges, T, work,
                      geom + GL.zero_begin, GL.zero_end - GL.zero_begin,
                      (flags & GRPG_FORWARD_NO_BACKWARD) == 0u,
                      drain_cnt_words ? (uint32_t*)(img + IL.ck_count) : (uint32_t*)tile_obj,
                      drain_cnt_words ? T + drain_cnt_words : tile_obj_words);   // (drained: flags and counters in one sweep)
    uint2* rects = hier ? (uint2*)(geom + GL.rects) : nullptr;
    uint2* rect_sorted = (uint2*)(geom + GL.rect_sorted);
    uint4* pre_counts = (uint4*)(geom + GL.pre_counts);
    if (segs == nullptr) {
      launch_preprocess(stream, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs,
                        cov3D_precomp, colors_precomp, cam, radii_int, rec_w, key_a, tiles, rects,
                        fat_sort ? ds_table : nullptr, pre_counts, marks);
    } else {
      // segment table: host structs -> pinned staging -> the geometry blob (asynchronous copy)
      if (!g_seg_staging)
        HIP_TRY(hipHostMalloc((void**)&g_seg_staging, sizeof(SegmentDev) * MAX_SEGMENTS, hipHostMallocDefault));
      uint32_t start = 0;
      for (int i = 0; i < nseg; i++) {
        SegmentDev& d = g_seg_staging[i];
        const grpg_model_segment& g = segs[i];
        d.xyz = g.xyz; d.scaling = g.scaling; d.rotation = g.rotation; d.opacity = g.opacity;
        d.fdc = g.features_dc; d.frest = g.features_rest; d.flip = g.flip;
        d.pad1 = (const void*)(uintptr_t)(layers ? ((layers->segment_class ? layers->segment_class[i] != 0
                                                                           : g.rigid != 0) ? 1u : 0u) : 0u);
        d.start = start; d.count = (uint32_t)g.count;
        d.fourier_dim = g.fourier_dim; d.rigid = g.rigid;
        for (int k = 0; k < 4; k++) d.rot[k] = g.obj_rot[k];
        for (int k = 0; k < 3; k++) d.trans[k] = g.obj_trans[k];
        d.pad0 = 0.f;
        for (int k = 0; k < MAX_FOURIER; k++) d.idft[k] = g.idft[k];
        start += (uint32_t)g.count;
      }
      SegmentDev* seg_dev = (SegmentDev*)(geom + GL.seg_table);
      HIP_TRY(hipMemcpyAsync(seg_dev, g_seg_staging, sizeof(SegmentDev) * (size_t)nseg,
                             hipMemcpyHostToDevice, stream));
      launch_preprocess_composed(stream, P, D, M, seg_dev, nseg, scale_modifier, cam, radii_int, rec_w,
                                 key_a, tiles, rects, fat_sort ? ds_table : nullptr, pre_counts, marks);
    }
    // num_rendered (and the coarse pair count) as sums of the per-Gaussian counts: in the pinned
    // word ~70 us into the frame, with an event behind it.  The host looks at it only after the
    // whole frame is enqueued (speculative mode) -- by then it has long arrived, so grpg_forward
    // returns the exact count without ever idling the stream or itself.
    // With the fat depth sort the sum rides in its first pass (one launch fewer; the count arrives
    // ~35 us later, still long before the host has enqueued the frame); the classic sort of very
    // large P: the separate one-workgroup launch right here.
#ifdef GRPG_PUBLISH_EARLY   // experiment build: the count by its own launch right behind preprocess
    const bool fold_publish = false;
#else
    const bool fold_publish = fat_sort;
#endif
    if (!fold_publish) {
      launch_publish_counts(stream, pre_counts, (uint32_t)((P + 255) / 256), pub_ptr, &gh->R_pre);
      HIP_TRY(hipEventRecord(pub_ev, stream));
    }
    STAGE_CHECK("preprocess");
    tm.mark(1);
    // (depth_bits, id) order: stable sort of ids by the 32-bit depth key.
    uint32_t* tiles_sorted = (uint32_t*)(geom + GL.tiles_sorted);
    const uint32_t* sorted_gid;
    // hierarchical binning on a grid of <= 255 x 255 tiles: the tile rectangles ride through the
    // depth sort as a packed payload and arrive in depth order together with the super-tile counts
    // (sort.hip RectPayload); larger grids: the coarse scan gathers them by sorted id instead
    const bool rect_sorted_by_sort = hier && cam.gx <= 255 && cam.gy <= 255;
    if (fat_sort) {   // drops the culled Gaussians: V pairs remain
      const uint32_t pnb = (uint32_t)((P + 255) / 256);
      uint32_t* key_c = (uint32_t*)(geom + GL.key_c);
      uint32_t* val_c = (uint32_t*)(geom + GL.val_c);
      if (rect_sorted_by_sort)   // scratch: rect_sorted's own first half, the not yet written offsets, aux_c
        depth_sort_fat(stream, (uint32_t)P, key_a, val_a, key_b, val_b, key_c, val_c, ds_table,
                       GL.nchunks_ds, &gh->V, &gh->key_base, with_pass3,
                       rects, (uint32_t*)rect_sorted, offsets,
                       (uint32_t*)(geom + GL.aux_c), rect_sorted, tiles_sorted, pre_counts, pnb, fold_publish,
                       pub_ptr, &gh->R_pre, pub_ev);
      else
        depth_sort_fat(stream, (uint32_t)P, key_a, val_a, key_b, val_b, key_c, val_c, ds_table,
                       GL.nchunks_ds, &gh->V, &gh->key_base, with_pass3,
                       nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, pre_counts, pnb, fold_publish, pub_ptr, &gh->R_pre, pub_ev);
      sorted_gid = val_a;
    } else {          // culled keys sort last (tile count 0); V stays P
      const bool in_b = radix_sort_pairs(stream, (uint32_t)P, nullptr, key_a, val_a, key_b, val_b, true,
                                         0, 32, table, totals, GL.nchunks_sort, false, tiles,
                                         tiles_sorted);
      sorted_gid = in_b ? val_b : val_a;  // 4 passes -> back in "a"
    }
    STAGE_CHECK("depth sort");
    tm.begin_tail();
    uint32_t* emit_win = (uint32_t*)(geom + GL.emit_win);
    // sort path: exclusive offsets of the per-Gaussian tile counts in depth order (the count itself
    // is already on its way to the host, see above)
    auto scan_tiles = [&]() -> int {
      tm.mark(2);
      launch_offsets_scan(stream, (uint32_t)P, &gh->V, tiles_sorted, fat_sort ? sorted_gid : nullptr, tiles,
                          offsets, block_sums, GL.nblocks_scan, &gh->R, nullptr, emit_win,
                          GL.emit_win_cap);
      STAGE_CHECK("offsets scan");
      return GRPG_OK;
    };
    auto render_tail = [&](char* binp, const BinLayout& L, const uint32_t* point_list, uint32_t cap,
                           bool classified) -> int {
      tm.mark(6);
      const CkptArgs ck = {(float*)(binp + L.total), (uint32_t*)(img + IL.ck_count),
                           (uint32_t*)(img + IL.bwd_ctl), (BlobHeader*)binp,
                           (uint32_t)(L.total / 256), ckpt_slots(cap)};
      if (layers)
        launch_render_layers(stream, ranges, const_cast<uint32_t*>(point_list), rec, width, height, cam.gx, cam.gy,
                             background, out_color, out_depth, out_alpha, work, frame_classes, cap,
                             classified, layers->layer_background, layers->out_color_bg,
                             layers->out_alpha_bg, layers->out_color_obj, layers->out_alpha_obj,
                             PCErr{&((BlobHeader*)img)->pc_timeout, hw->dev_ptr + 3}, epi);
      else
      launch_render_forward(stream, ranges, point_list, rec, width, height, cam.gx, cam.gy, background,
                            out_color, out_depth, out_alpha, n_contrib, work, frame_classes, cap,
                            (flags & GRPG_FORWARD_NO_BACKWARD) == 0u, classified,
                            with_ckpt ? &ck : nullptr,
                            PCErr{&((BlobHeader*)img)->pc_timeout, hw->dev_ptr + 3},
                            S > 0 ? semantics : nullptr, S, out_semantic, epi);
      STAGE_CHECK("render");
      if (debug)   // the stage check has synchronised: a hand-over timeout of THIS frame is known
        if (int rc = check_async_error(hw)) return rc;
      tm.mark(7);
      if (S > RENDER_NSEM) {   // the first RENDER_NSEM planes rode in the render launch
        launch_render_semantic(stream, ranges, point_list, rec, semantics, S, RENDER_NSEM, width, height,
                               cam.gx, cam.gy, out_semantic);
        STAGE_CHECK("semantic render");
      }
      tm.mark(-1);
      return GRPG_OK;
    };

    const int tbits = bits_for(T);
    const int passes = radix_sort_num_passes(0, tbits);
    const int bits0 = radix_sort_first_pass_bits(0, tbits);
    // sort path: everything behind the offsets scan, for a binning blob of capacity `cap`
    auto tail_sort = [&](char* binp, const BinLayout& L, uint32_t cap) -> int {
      uint32_t* bkey_a = (uint32_t*)(binp + L.key_a);
      uint32_t* bkey_b = (uint32_t*)(binp + L.key_b);
      uint32_t* bval_a = (uint32_t*)(binp + L.val_a);
      uint32_t* bval_b = (uint32_t*)(binp + L.val_b);
      uint32_t* btable = (uint32_t*)(binp + L.table);
      uint32_t* btotals = (uint32_t*)(binp + L.totals);
      tm.mark(3);
      // emit into whichever pair makes the LAST pass land in "a" (point_list lives in val_a)
      const bool start_in_b = (passes & 1) != 0;
      launch_emit(stream, &gh->V, &gh->R, cap, sorted_gid, offsets, emit_win, GL.emit_win_cap, rec,
                  cam.gx, cam.gy, start_in_b ? bkey_b : bkey_a, start_in_b ? bval_b : bval_a,
                  passes > 0 ? btable : nullptr, (1u << bits0) - 1u, L.nchunks_sort);
      STAGE_CHECK("emit");
      tm.mark(4);
      if (passes > 0 && cap > 0) {
        bool res_b;
        if (start_in_b)
          res_b = !radix_sort_pairs(stream, cap, &gh->R, bkey_b, bval_b, bkey_a, bval_a, false, 0, tbits,
                                    btable, btotals, L.nchunks_sort, true);
        else
          res_b = radix_sort_pairs(stream, cap, &gh->R, bkey_a, bval_a, bkey_b, bval_b, false, 0, tbits,
                                   btable, btotals, L.nchunks_sort, true);
        if (res_b) return fail(GRPG_ERR_HIP, "internal: tile sort landed in the wrong buffer");
      }
      STAGE_CHECK("tile sort");
      tm.mark(5);
      launch_tile_ranges(stream, &gh->R, cap, bkey_a, ranges, T);
      STAGE_CHECK("tile ranges");
      return render_tail(binp, L, bval_a, cap, false);
    };
    // hierarchical path (hier_binning.hip): everything behind the depth sort.  Stage slots: 2 = scan
    // over the super-tile counts, 3 = coarse emit, 4 = coarse partition + super-tile runs,
    // 5 = segment counts, tile ranges and the point-list fill.
    const int sgx = (cam.gx + STILE - 1) / STILE, sgy = (cam.gy + STILE - 1) / STILE;
    const int cbits = bits_for(NS);
    const int cpasses = radix_sort_num_passes(0, cbits);
    const int cbits0 = radix_sort_first_pass_bits(0, cbits);
    auto tail_hier = [&](char* binp, const BinLayout& L, uint32_t cap, uint32_t ccap) -> int {
      uint32_t* ckey_a = (uint32_t*)(binp + L.key_a);
      uint32_t* ckey_b = (uint32_t*)(binp + L.key_b);
      uint32_t* cval_a = (uint32_t*)(binp + L.val_b);
      uint32_t* cval_b = (uint32_t*)(binp + L.val_c);
      uint32_t* plist = (uint32_t*)(binp + L.val_a);
      uint32_t* btable = (uint32_t*)(binp + L.table);
      uint32_t* btotals = (uint32_t*)(binp + L.totals);
      uint2* cranges = (uint2*)(binp + L.cranges);
      tm.mark(2);
      // counts and rectangles already in depth order (from the sort's last pass): the emit scans the offsets it
      // needs itself from the block sums (binning.hip emit_coarse_fused_kernel); otherwise the two-launch scan
      const bool fused_emit = rect_sorted_by_sort && emit_coarse_fused_ok(GL.nblocks_scan) && !UNFUSED_COARSE_EMIT;
      if (fused_emit)
        launch_offsets_reduce(stream, (uint32_t)P, &gh->V, tiles_sorted, block_sums, GL.nblocks_scan);
      else if (rect_sorted_by_sort)
        launch_offsets_scan(stream, (uint32_t)P, &gh->V, tiles_sorted, nullptr, nullptr, offsets,
                            block_sums, GL.nblocks_scan, &gh->Rc, nullptr, emit_win, GL.emit_win_cap);
      else
        launch_offsets_scan(stream, (uint32_t)P, &gh->V, tiles_sorted, sorted_gid, nullptr, offsets,
                            block_sums, GL.nblocks_scan, &gh->Rc, nullptr, emit_win, GL.emit_win_cap, rects,
                            rect_sorted);
      STAGE_CHECK("coarse offsets scan");
      tm.mark(3);
      if (fused_emit)
        launch_emit_coarse_fused(stream, &gh->V, &gh->Rc, ccap, sorted_gid, tiles_sorted, block_sums,
                                 GL.nblocks_scan, rect_sorted, sgx, ckey_a, cval_a, cpasses > 0 ? btable : nullptr,
                                 (1u << cbits0) - 1u, L.nchunks_coarse, cranges, NS);
      else
        launch_emit_coarse(stream, &gh->V, &gh->Rc, ccap, sorted_gid, rect_sorted, offsets, emit_win,
                           GL.emit_win_cap, sgx, sgy, ckey_a, cval_a, cpasses > 0 ? btable : nullptr,
                           (1u << cbits0) - 1u, L.nchunks_coarse, cranges, NS);
      STAGE_CHECK("coarse emit");
      tm.mark(4);
      bool in_b = false;
      if (cpasses > 0 && ccap > 0)
        in_b = radix_sort_pairs(stream, ccap, &gh->Rc, ckey_a, cval_a, ckey_b, cval_b, false, 0, cbits,
                                btable, btotals, L.nchunks_coarse, true);
      STAGE_CHECK("coarse partition");
      const uint32_t* ckey = in_b ? ckey_b : ckey_a;
      // one pass (<= 256 super-tiles): its digit totals ARE the run lengths; otherwise find the runs
      const bool runs_from_totals = cpasses == 1 && ccap > 0;
      if (!runs_from_totals) launch_tile_ranges(stream, &gh->Rc, ccap, ckey, cranges, NS, 0xFFFFu);
      STAGE_CHECK("super-tile runs");
      tm.mark(5);
      const uint32_t* cval = in_b ? cval_b : cval_a;
      launch_hier_count(stream, cranges, runs_from_totals ? btotals : nullptr, NS, binp + L.seg_desc, (uint2*)(binp + L.st_seg),
                        (uint32_t*)(binp + L.nseg), L.max_seg, ckey, cam.gx, cam.gy,
                        (uint32_t*)(binp + L.seg_table), (uint32_t*)(binp + L.tile_tot),
                        (uint32_t*)(binp + L.tile_start), ranges, &gh->R, nullptr,
                        &gh->Rc, (BlobHeader*)binp, cap, ccap, work, frame_classes);
      STAGE_CHECK("tile counts");
      launch_hier_fill(stream, binp + L.seg_desc, (const uint32_t*)(binp + L.nseg), L.max_seg, ckey, cval, rec,
                       cam.gx, cam.gy, (const uint32_t*)(binp + L.seg_table),
                       (const uint32_t*)(binp + L.tile_start), cap, plist);
      STAGE_CHECK("point list fill");
      return render_tail(binp, L, plist, cap, true);
    };
    auto carve = [&](uint32_t cap, uint32_t ccap, bool rezero) -> int {
      Rcap = cap;
      Ccap = ccap;
      BL = layout_for(Rcap, Ccap);
      bin = binning_alloc(blob_bytes(BL, Rcap), binning_user);
      if (!bin) return fail(GRPG_ERR_ALLOC, "binning buffer allocation failed");
      launch_bin_header(stream, bin, (uint32_t)P, Rcap, (uint32_t)width, (uint32_t)height, (uint32_t)S,
                        rezero ? ranges : nullptr, T, rezero ? work : nullptr);
      return GRPG_OK;
    };
    const bool spec_mode = g_binning_mode.load() == GRPG_BINNING_SPECULATIVE;

    // Speculative: the whole frame is enqueued for the remembered capacities, then the host reads
    // the count (published right behind preprocess: normally there already).  No history / exact
    // mode: read the count first, carve the blob for it, then enqueue the tail -- like the
    // reference's one blocking read (rasterizer_impl.cu:284), only much earlier in the frame.
    uint32_t Rc_seen = 0u;   // coarse count of this frame, once the host has seen it
    auto run_tail = [&]() -> int {
      if (hier) return tail_hier(bin, BL, Rcap, Ccap);
      if (int rc = scan_tiles()) return rc;
      return tail_sort(bin, BL, Rcap);
    };
    if (speculative) {
      if (int rc = run_tail()) return rc;
      if (defer) {
        // deferred frame: the count is checked by grpg_frame_status, nothing to wait for here
        defer->state = 1;
        defer->Rcap = Rcap; defer->Ccap = Ccap; defer->hier = hier; defer->key = ck;
        defer->with_pass3 = with_pass3 || !fat_sort;
        tm.finish();
        return 0;
      }
    }
    HIP_TRY(hipEventSynchronize(pub_ev));
    R = (defer ? defer->host_ptr : hw->host_ptr)[0];
    if (hier) Rc_seen = (defer ? defer->host_ptr : hw->host_ptr)[1];
    const bool far_seen = fat_sort && (defer ? defer->host_ptr : hw->host_ptr)[2] != 0u;
    if (R > 0x7FFFFFFFu) return fail(GRPG_ERR_INVALID_ARGUMENT, "num_rendered exceeds int32");
    if (far_seen && !with_pass3) {
      // the scene's depth range grew beyond what three sort passes order: this frame's lists are in
      // the wrong order.  Remember it and render the frame again, this time with the fourth pass.
      update_hint(ck, R, Rc_seen, true);
      return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user,
                          P, D, M, S, background, width, height, means3D, shs, colors_precomp, semantics,
                          opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                          cam_pos, tan_fovx, tan_fovy, out_color, out_depth, out_alpha, out_semantic, radii,
                          debug, hip_stream, segs, nseg, flags, defer, true, layers, epi);
    }
    if (!speculative || R > Rcap || (hier && Rc_seen > Ccap)) {
      // first frame of a shape / exact mode: carve for the count just read.  Capacity overflow: the
      // speculative tail clamped its work to the old capacities and its results are discarded; both
      // counts are exact, so one redo on a larger blob settles it.
      const bool redo = speculative;
      uint32_t cap = Rcap, ccap = Ccap;
      if (!speculative) {
        cap = spec_mode ? padded_capacity(R) : R;
        ccap = hier ? (spec_mode ? padded_capacity(Rc_seen) : Rc_seen) : cap;
      } else {
        if (R > Rcap) cap = padded_capacity(R);
        if (hier && Rc_seen > Ccap) ccap = padded_capacity(Rc_seen);
      }
      if (ccap > cap) ccap = cap;     // a (Gaussian, super-tile) pair holds >= 1 instance
      if (redo) tm.restart_tail();
      if (int rc = carve(cap, ccap, redo)) return rc;
      if (int rc = run_tail()) return rc;
    }
    if (R > 0x7FFFFFFFu) return fail(GRPG_ERR_INVALID_ARGUMENT, "num_rendered exceeds int32");
    update_hint(ck, R, Rc_seen, far_seen);
    tm.finish();
    if (defer) { defer->state = 2; defer->result = (int)R; }   // no capacity history: ran synchronously
  } else {
    // P == 0: the reference launches nothing and its pre-zeroed planes stay zero
    // (rasterize_points.cu:85-86,123); write the zeros explicitly.
    if (out_color) HIP_TRY(hipMemsetAsync(out_color, 0, 3 * N * 4, stream));
    if (out_depth) HIP_TRY(hipMemsetAsync(out_depth, 0, N * 4, stream));
    if (out_alpha) HIP_TRY(hipMemsetAsync(out_alpha, 0, N * 4, stream));
    if (epi) launch_frame_epilogue_empty(stream, width, height, *epi, out_color);   // sky over nothing, bytes
    if (S > 0 && out_semantic) HIP_TRY(hipMemsetAsync(out_semantic, 0, (size_t)S * N * 4, stream));
    HIP_TRY(hipMemsetAsync(n_contrib, 0, N * 4, stream));
    if (layers) {
      // ONE contract for an empty layer: its colour plane is layer_background, its alpha plane zero --
      // what the kernels write for an empty class when P > 0, and what the reference's render_kernel returns for a
      // model set without Gaussians (street_gaussian_renderer.py:131-144: ones on a white background, acc zeros)
      launch_fill_layer_planes(stream, N, layers->layer_background, layers->out_color_bg, layers->out_alpha_bg,
                               layers->out_color_obj, layers->out_alpha_obj);
    }
    char* bin = binning_alloc(bin_layout(0).total, binning_user);
    if (!bin) return fail(GRPG_ERR_ALLOC, "binning buffer allocation failed");
    launch_frame_init(stream, geom, bin, img, 0u, 0u, 0u, (uint32_t)width, (uint32_t)height,
                      (uint32_t)S, ranges, T, work, geom + GL.zero_begin, GL.zero_end - GL.zero_begin,
                      (flags & GRPG_FORWARD_NO_BACKWARD) == 0u);
  }
  if (defer && defer->state == 0) { defer->state = 2; defer->result = (int)R; }
  return (int)R;
}

// ---- deferred frames: slot management ----
int defer_resolve(DeferSlot& d, bool wait) {
  if (d.state == 2) return d.result;
  if (d.state != 1) return fail(GRPG_ERR_INVALID_ARGUMENT, "ticket does not name a frame in flight");
  if (wait) {
    HIP_TRY(hipEventSynchronize(d.ev));
  } else {
    const hipError_t q = hipEventQuery(d.ev);
    if (q == hipErrorNotReady) return GRPG_ERR_NOT_READY;
    if (q != hipSuccess) return fail(GRPG_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
  }
  const uint32_t R = d.host_ptr[0], Rc = d.hier ? d.host_ptr[1] : 0u;
  const bool far = d.host_ptr[2] != 0u;
  // a clamped coarse list undercounts R; either overflow invalidates the frame's outputs -- and so
  // does a depth range that needed the fourth sort pass the frame was enqueued without
  const bool ok = R <= d.Rcap && (!d.hier || Rc <= d.Ccap) && R <= 0x7FFFFFFFu && (!far || d.with_pass3);
  update_hint(d.key, R > Rc ? R : Rc, Rc, far);   // (an undercounted R is at least the coarse count)
  d.state = 2;
  d.result = ok ? (int)R : GRPG_ERR_CAPACITY;
  return d.result;
}

DeferSlot* defer_acquire(int* ticket) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const int idx = g_defer_next;
  g_defer_next = (g_defer_next + 1) % DEFER_SLOTS;
  DeferSlot& d = g_defer[idx];
  if (d.state == 1) (void)defer_resolve(d, true);   // comes round while pending: look at it now
  if (d.host_ptr && d.dev != dev) {                  // pinned words map into one device's space
    (void)hipHostFree(d.host_ptr);
    (void)hipEventDestroy(d.ev);
    const int gen = d.generation;
    d = DeferSlot{};
    d.generation = gen;
  }
  if (!d.host_ptr) {
    if (hipHostMalloc((void**)&d.host_ptr, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
    if (hipHostGetDevicePointer((void**)&d.dev_ptr, d.host_ptr, 0) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    d.dev = dev;
  }
  d.host_ptr[0] = 0u; d.host_ptr[1] = 0u; d.host_ptr[2] = 0u;
  d.state = 0; d.result = 0;
  d.generation = (d.generation + 1) & 0xFFFFF;
  *ticket = d.generation * DEFER_SLOTS + idx;
  return &d;
}

int check_segments(const grpg_model_segment* segs, int nseg, int M, long long* P_out) {
  if (!segs || nseg <= 0 || nseg > GRPG_MAX_SEGMENTS)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "need 1..GRPG_MAX_SEGMENTS segments");
  long long P = 0;
  for (int i = 0; i < nseg; i++) {
    const grpg_model_segment& g = segs[i];
    if (g.count <= 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "segment with count <= 0");
    if (!g.xyz || !g.scaling || !g.rotation || !g.opacity || !g.features_dc || (M > 1 && !g.features_rest))
      return fail(GRPG_ERR_INVALID_ARGUMENT, "segment with a NULL parameter array");
    if (g.fourier_dim < 1 || g.fourier_dim > GRPG_MAX_FOURIER)
      return fail(GRPG_ERR_INVALID_ARGUMENT, "fourier_dim must be in 1..GRPG_MAX_FOURIER");
    P += g.count;
  }
  if (P > (long long)ID_MASK) return fail(GRPG_ERR_INVALID_ARGUMENT, "P must be < 2^28");
  *P_out = P;
  return GRPG_OK;
}

}  // namespace

extern "C" {

int grpg_forward(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                 void* binning_user, grpg_alloc_fn image_alloc, void* image_user, int P, int D,
                 int M, int S, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* semantics, const float* opacities, const float* scales,
                 float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                 float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                 float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug,
                 void* hip_stream) {
  (void)prefiltered;  // reference: __trap() on a culled point when set; always False in practice
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                      image_user, P, D, M, S, background, width, height, means3D, shs, colors_precomp,
                      semantics, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                      viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, out_color, out_depth,
                      out_alpha, out_semantic, radii, debug, hip_stream, nullptr, 0);
}

int grpg_forward_flags(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                       void* binning_user, grpg_alloc_fn image_alloc, void* image_user, int P, int D,
                       int M, int S, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* semantics, const float* opacities, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                       float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug,
                       void* hip_stream, unsigned flags) {
  (void)prefiltered;
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                      image_user, P, D, M, S, background, width, height, means3D, shs, colors_precomp,
                      semantics, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                      viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, out_color, out_depth,
                      out_alpha, out_semantic, radii, debug, hip_stream, nullptr, 0, flags);
}

int grpg_forward_layers(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                        void* binning_user, grpg_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                        const float* background, int width, int height, const float* means3D,
                        const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                        const float* cam_pos, float tan_fovx, float tan_fovy,
                        const unsigned char* layer_class, const float* layer_background, float* out_color,
                        float* out_depth, float* out_alpha, float* out_color_bg, float* out_alpha_bg,
                        float* out_color_obj, float* out_alpha_obj, int* radii, int debug, void* hip_stream) {
  g_last_error.clear();
  if (!layer_background || !out_color_bg || !out_alpha_bg || !out_color_obj || !out_alpha_obj)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL layer background / layer plane pointer");
  if (P > 0 && !layer_class) return fail(GRPG_ERR_INVALID_ARGUMENT, "layer_class NULL");
  if ((unsigned)P >= (1u << 27))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "a layered frame needs P < 2^27 (the class travels in bit 27 of the point list)");
  const LayerArgs la = {nullptr, layer_class, layer_background, out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj};
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D,
                      M, 0, background, width, height, means3D, shs, colors_precomp, nullptr, opacities, scales,
                      scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                      tan_fovy, out_color, out_depth, out_alpha, nullptr, radii, debug, hip_stream, nullptr, 0,
                      GRPG_FORWARD_NO_BACKWARD, nullptr, false, &la);
}

int grpg_forward_deferred(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                          void* binning_user, grpg_alloc_fn image_alloc, void* image_user, int P, int D,
                          int M, int S, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* semantics, const float* opacities, const float* scales,
                          float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                          float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                          float* out_depth, float* out_alpha, float* out_semantic, int* radii, int debug,
                          void* hip_stream, unsigned flags, int* ticket) {
  (void)prefiltered;
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (!ticket) return fail(GRPG_ERR_INVALID_ARGUMENT, "ticket must not be NULL");
  DeferSlot* d = defer_acquire(ticket);
  if (!d) return fail(GRPG_ERR_HIP, "pinned status words / event allocation failed");
  const int rc = forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                              image_user, P, D, M, S, background, width, height, means3D, shs,
                              colors_precomp, semantics, opacities, scales, scale_modifier, rotations,
                              cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                              out_color, out_depth, out_alpha, out_semantic, radii, debug, hip_stream,
                              nullptr, 0, flags, d);
  if (rc < 0) { d->state = 2; d->result = rc; return rc; }
  return GRPG_OK;
}

int grpg_frame_status(int ticket, int wait, int* num_rendered) {
  g_last_error.clear();
  if (ticket < 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "no such ticket");
  DeferSlot& slot = g_defer[ticket % DEFER_SLOTS];
  if (slot.generation != ticket / DEFER_SLOTS || slot.state == 0)
    return fail(GRPG_ERR_INVALID_ARGUMENT,
                "no such ticket on this thread: tickets are valid only on the thread that enqueued the "
                "frame, and only until GRPG_MAX_DEFERRED_FRAMES (64) later frames have been enqueued there");
  for (auto& hw : g_host_words)
    if (int rc = check_async_error(&hw)) return rc;
  const int r = defer_resolve(slot, wait != 0);
  if (r >= 0) { if (num_rendered) *num_rendered = r; return GRPG_OK; }
  if (r == GRPG_ERR_CAPACITY)
    g_last_error = "the frame overflowed the capacity it was enqueued with; render it again";
  return r;
}

int grpg_forward_composed_flags(grpg_alloc_fn geometry_alloc, void* geometry_user,
                                grpg_alloc_fn binning_alloc, void* binning_user,
                                grpg_alloc_fn image_alloc, void* image_user,
                                const grpg_model_segment* segments, int num_segments, int D, int M,
                                const float* background, int width, int height, float scale_modifier,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                float tan_fovx, float tan_fovy, float* out_color, float* out_depth,
                                float* out_alpha, int* radii, int debug, void* hip_stream, unsigned flags) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  long long P = 0;
  if (int rc = check_segments(segments, num_segments, M, &P)) return rc;
  if (M < 1 || M > 16 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "SH degree needs (D+1)^2 <= M <= 16");
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                      image_user, (int)P, D, M, 0, background, width, height, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, scale_modifier, nullptr, nullptr, viewmatrix,
                      projmatrix, cam_pos, tan_fovx, tan_fovy, out_color, out_depth, out_alpha, nullptr,
                      radii, debug, hip_stream, segments, num_segments, flags);
}

int grpg_forward_composed_layers(grpg_alloc_fn geometry_alloc, void* geometry_user,
                                 grpg_alloc_fn binning_alloc, void* binning_user,
                                 grpg_alloc_fn image_alloc, void* image_user,
                                 const grpg_model_segment* segments, int num_segments,
                                 const unsigned char* segment_class, int D, int M,
                                 const float* background, const float* layer_background, int width, int height,
                                 float scale_modifier, const float* viewmatrix, const float* projmatrix,
                                 const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color,
                                 float* out_depth, float* out_alpha, float* out_color_bg, float* out_alpha_bg,
                                 float* out_color_obj, float* out_alpha_obj, int* radii, int debug,
                                 void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  long long P = 0;
  if (int rc = check_segments(segments, num_segments, M, &P)) return rc;
  if (M < 1 || M > 16 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "SH degree needs (D+1)^2 <= M <= 16");
  if (!layer_background || !out_color_bg || !out_alpha_bg || !out_color_obj || !out_alpha_obj)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL layer background / layer plane pointer");
  if (P >= (1ll << 27))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "a layered frame needs P < 2^27 (the class travels in bit 27 of the point list)");
  const LayerArgs la = {segment_class, nullptr, layer_background, out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj};
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                      image_user, (int)P, D, M, 0, background, width, height, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, scale_modifier, nullptr, nullptr, viewmatrix,
                      projmatrix, cam_pos, tan_fovx, tan_fovy, out_color, out_depth, out_alpha, nullptr,
                      radii, debug, hip_stream, segments, num_segments, GRPG_FORWARD_NO_BACKWARD, nullptr, false, &la);
}

// grpg_frame_epilogue -> FrameEpilogue (pointers checked); returns GRPG_OK or fails
static int resolve_epilogue(const grpg_frame_epilogue* e, bool planes, FrameEpilogue* out) {
  if (e->sky_cube != nullptr && (e->sky_res <= 0 || e->ray_matrix == nullptr))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "frame epilogue: the sky composite needs sky_res > 0 and a ray matrix");
  out->sky_cube = e->sky_cube; out->sky_res = e->sky_res;
  out->ray_m_dev = (e->sky_cube && e->ray_matrix_on_device) ? e->ray_matrix : nullptr;
  for (int i = 0; i < 9; i++) out->ray_m[i] = (e->sky_cube && !e->ray_matrix_on_device) ? e->ray_matrix[i] : 0.f;
  out->sky_fill = e->sky_fill; out->clamp = e->clamp; out->rgb8 = e->out_rgb8; out->truncate = e->truncate;
  out->rgb8_host = (e->out_rgb8 != nullptr && e->out_rgb8_on_host) ? 1 : 0;
  out->planes = planes ? 1 : 0;
  return GRPG_OK;
}

// layer planes: all four or none
static int layer_planes_given(const float* a, const float* b, const float* c, const float* d) {
  const int n = (a != nullptr) + (b != nullptr) + (c != nullptr) + (d != nullptr);
  return n == 4 ? 1 : (n == 0 ? 0 : -1);
}

int grpg_forward_frame(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                       void* binning_user, grpg_alloc_fn image_alloc, void* image_user, int P, int D, int M,
                       const float* background, int width, int height, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                       float tan_fovy, const unsigned char* layer_class, const float* layer_background,
                       float* out_color, float* out_depth, float* out_alpha, float* out_color_bg,
                       float* out_alpha_bg, float* out_color_obj, float* out_alpha_obj, int* radii, int debug,
                       void* hip_stream, const grpg_frame_epilogue* epilogue) {
  g_last_error.clear();
  const int lay = layer_planes_given(out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj);
  if (lay < 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "layer planes: give all four or none");
  if (lay && (!layer_background || (P > 0 && !layer_class)))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "a layered frame needs layer_class and layer_background");
  if (lay && (unsigned)P >= (1u << 27))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "a layered frame needs P < 2^27 (the class travels in bit 27 of the point list)");
  const bool planes = out_color && out_depth && out_alpha;
  FrameEpilogue fe{};
  if (epilogue) if (int rc = resolve_epilogue(epilogue, planes, &fe)) return rc;
  const LayerArgs la = {nullptr, layer_class, layer_background, out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj};
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc, image_user, P, D,
                      M, 0, background, width, height, means3D, shs, colors_precomp, nullptr, opacities, scales,
                      scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx,
                      tan_fovy, out_color, out_depth, out_alpha, nullptr, radii, debug, hip_stream, nullptr, 0,
                      GRPG_FORWARD_NO_BACKWARD, nullptr, false, lay ? &la : nullptr, epilogue ? &fe : nullptr);
}

int grpg_forward_composed_frame(grpg_alloc_fn geometry_alloc, void* geometry_user, grpg_alloc_fn binning_alloc,
                                void* binning_user, grpg_alloc_fn image_alloc, void* image_user,
                                const grpg_model_segment* segments, int num_segments,
                                const unsigned char* segment_class, int D, int M, const float* background,
                                const float* layer_background, int width, int height, float scale_modifier,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                float tan_fovx, float tan_fovy, float* out_color, float* out_depth,
                                float* out_alpha, float* out_color_bg, float* out_alpha_bg, float* out_color_obj,
                                float* out_alpha_obj, int* radii, int debug, void* hip_stream,
                                const grpg_frame_epilogue* epilogue) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  long long P = 0;
  if (int rc = check_segments(segments, num_segments, M, &P)) return rc;
  if (M < 1 || M > 16 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "SH degree needs (D+1)^2 <= M <= 16");
  const int lay = layer_planes_given(out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj);
  if (lay < 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "layer planes: give all four or none");
  if (lay && !layer_background) return fail(GRPG_ERR_INVALID_ARGUMENT, "a layered frame needs layer_background");
  if (lay && P >= (1ll << 27))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "a layered frame needs P < 2^27 (the class travels in bit 27 of the point list)");
  const bool planes = out_color && out_depth && out_alpha;
  FrameEpilogue fe{};
  if (epilogue) if (int rc = resolve_epilogue(epilogue, planes, &fe)) return rc;
  const LayerArgs la = {segment_class, nullptr, layer_background, out_color_bg, out_alpha_bg, out_color_obj, out_alpha_obj};
  return forward_impl(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                      image_user, (int)P, D, M, 0, background, width, height, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, scale_modifier, nullptr, nullptr, viewmatrix,
                      projmatrix, cam_pos, tan_fovx, tan_fovy, out_color, out_depth, out_alpha, nullptr,
                      radii, debug, hip_stream, segments, num_segments, GRPG_FORWARD_NO_BACKWARD, nullptr, false,
                      lay ? &la : nullptr, epilogue ? &fe : nullptr);
}

int grpg_forward_composed(grpg_alloc_fn geometry_alloc, void* geometry_user,
                          grpg_alloc_fn binning_alloc, void* binning_user,
                          grpg_alloc_fn image_alloc, void* image_user,
                          const grpg_model_segment* segments, int num_segments, int D, int M,
                          const float* background, int width, int height, float scale_modifier,
                          const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                          float tan_fovx, float tan_fovy, float* out_color, float* out_depth,
                          float* out_alpha, int* radii, int debug, void* hip_stream) {
  return grpg_forward_composed_flags(geometry_alloc, geometry_user, binning_alloc, binning_user, image_alloc,
                                     image_user, segments, num_segments, D, M, background, width, height,
                                     scale_modifier, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                                     out_color, out_depth, out_alpha, radii, debug, hip_stream,
                                     GRPG_FORWARD_NO_BACKWARD);
}

int grpg_backward_composed(const grpg_model_segment* segments, const grpg_model_segment_grad* grads,
                           int num_segments, int D, int M, int R, const float* background, int width,
                           int height, float scale_modifier, const float* viewmatrix,
                           const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                           const int* radii, const float* alphas, char* geom_buffer, char* binning_buffer,
                           char* image_buffer, const float* dL_dpix, const float* dL_dpix_depth,
                           const float* dL_dalphas, float* dL_dmean2D, float* dL_dposes, int debug,
                           void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  long long Pll = 0;
  if (int rc = check_segments(segments, num_segments, M, &Pll)) return rc;
  const int P = (int)Pll;
  if (M < 1 || M > 16 || D < 0 || D > 3 || (D + 1) * (D + 1) > M)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "SH degree needs (D+1)^2 <= M <= 16");
  if (!grads || !geom_buffer || !binning_buffer || !image_buffer)
    return fail(GRPG_ERR_BAD_BUFFER, "NULL gradient table / state buffer");
  if (int rc = geom_check_backward(geom_buffer, P, (hipStream_t)hip_stream)) return rc;
  for (auto& hw : g_host_words)
    if (int rc = check_async_error(&hw)) return rc;
  if (!dL_dpix || !dL_dpix_depth || !dL_dalphas || !alphas || !dL_dmean2D || !dL_dposes || !radii ||
      !background || !viewmatrix || !projmatrix || !campos)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL pointer");
  for (int i = 0; i < num_segments; i++) {
    const grpg_model_segment_grad& g = grads[i];
    if (!g.xyz || !g.scaling || !g.rotation || !g.opacity || !g.features_dc || (M > 1 && !g.features_rest))
      return fail(GRPG_ERR_INVALID_ARGUMENT, "segment gradient with a NULL array");
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  const CameraArgs cam = make_camera(viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy);
  const uint32_t T = (uint32_t)cam.gx * (uint32_t)cam.gy;
  const GeomLayout GL = geom_layout((size_t)P);
  const ImgLayout IL = img_layout(T, (size_t)width * height);
  const RecView rec = {(const float4*)(geom_buffer + GL.rec)};
  const uint32_t* point_list = (const uint32_t*)(binning_buffer + bin_layout(0).val_a);
  const uint2* ranges = (const uint2*)(image_buffer + IL.ranges);
  const uint32_t* n_contrib = (const uint32_t*)(image_buffer + IL.n_contrib);
  // tables: the segments (as in the forward; re-uploaded: the caller may hand over other arrays of
  // the same values) and the six output pointers per segment, through pinned staging
  BwdStagingSlot* stg = bwd_staging_acquire();
  if (!stg) return fail(GRPG_ERR_HIP, "pinned staging allocation failed");
  uint32_t start = 0;
  for (int i = 0; i < num_segments; i++) {
    SegmentDev& d = stg->segs[i];
    const grpg_model_segment& g = segments[i];
    d.xyz = g.xyz; d.scaling = g.scaling; d.rotation = g.rotation; d.opacity = g.opacity;
    d.fdc = g.features_dc; d.frest = g.features_rest; d.flip = g.flip; d.pad1 = nullptr;
    d.start = start; d.count = (uint32_t)g.count;
    d.fourier_dim = g.fourier_dim; d.rigid = g.rigid;
    for (int k = 0; k < 4; k++) d.rot[k] = g.obj_rot[k];
    for (int k = 0; k < 3; k++) d.trans[k] = g.obj_trans[k];
    d.pad0 = 0.f;
    for (int k = 0; k < MAX_FOURIER; k++) d.idft[k] = g.idft[k];
    start += (uint32_t)g.count;
    stg->grads[i] = grads[i];
  }
  SegmentDev* seg_dev = (SegmentDev*)(geom_buffer + GL.seg_table);
  void* seg_grad_dev = (void*)(geom_buffer + GL.seg_grad_table);
  HIP_TRY(hipMemcpyAsync(seg_dev, stg->segs, sizeof(SegmentDev) * (size_t)num_segments,
                         hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(seg_grad_dev, stg->grads, sizeof(grpg_model_segment_grad) * (size_t)num_segments,
                         hipMemcpyHostToDevice, stream));
  HIP_TRY(hipEventRecord(stg->ev, stream));
  float* grad_rec = (float*)(geom_buffer + GL.grad_rec);
  HIP_TRY(hipMemsetAsync(grad_rec, 0, (size_t)P * GRAD_STRIDE * sizeof(float), stream));
  BwdTimingRecord* const btr = bwd_timing_begin(stream);
  BwdTimingGuard btg(btr);
  launch_render_backward(stream, ranges, point_list, rec, nullptr, 0, width, height, cam.gx, cam.gy,
                         background, alphas, n_contrib, (const uint32_t*)(image_buffer + IL.work), dL_dpix,
                         dL_dpix_depth, dL_dalphas, nullptr, grad_rec, nullptr,
                         (const BlobHeader*)binning_buffer, (const uint32_t*)(image_buffer + IL.ck_count),
                         (const uint32_t*)(image_buffer + IL.bwd_ctl), (uint32_t)R);
  STAGE_CHECK("render backward");
  bwd_timing_mark(btr, 1, stream);
  launch_preprocess_backward_composed(stream, P, D, M, seg_dev, seg_grad_dev, num_segments, radii, rec,
                                      scale_modifier, cam, grad_rec, dL_dmean2D,
                                      (float*)(geom_buffer + GL.pose_acc), dL_dposes);
  bwd_timing_mark(btr, 2, stream);
  btg.done = true;
  STAGE_CHECK("preprocess backward (composed)");
  return GRPG_OK;
}

int grpg_compose(const grpg_model_segment* segments, int num_segments, int M, float* means3D,
                 float* scales, float* rotations, float* opacities, float* shs, void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  long long P = 0;
  if (int rc = check_segments(segments, num_segments, M, &P)) return rc;
  if (M < 1 || M > 16) return fail(GRPG_ERR_INVALID_ARGUMENT, "M must be in 1..16");
  if (!means3D || !scales || !rotations || !opacities || !shs)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL output pointer");
  hipStream_t stream = (hipStream_t)hip_stream;
  // standalone call: the table travels through a temporary device buffer (stream-ordered free)
  std::vector<SegmentDev> host((size_t)num_segments);
  uint32_t start = 0;
  for (int i = 0; i < num_segments; i++) {
    SegmentDev& d = host[(size_t)i];
    const grpg_model_segment& g = segments[i];
    d.xyz = g.xyz; d.scaling = g.scaling; d.rotation = g.rotation; d.opacity = g.opacity;
    d.fdc = g.features_dc; d.frest = g.features_rest; d.flip = g.flip; d.pad1 = nullptr;
    d.start = start; d.count = (uint32_t)g.count;
    d.fourier_dim = g.fourier_dim; d.rigid = g.rigid;
    for (int k = 0; k < 4; k++) d.rot[k] = g.obj_rot[k];
    for (int k = 0; k < 3; k++) d.trans[k] = g.obj_trans[k];
    d.pad0 = 0.f;
    for (int k = 0; k < MAX_FOURIER; k++) d.idft[k] = g.idft[k];
    start += (uint32_t)g.count;
  }
  SegmentDev* dev_tab = nullptr;
  HIP_TRY(hipMalloc((void**)&dev_tab, sizeof(SegmentDev) * host.size()));
  hipError_t e = hipMemcpyAsync(dev_tab, host.data(), sizeof(SegmentDev) * host.size(),
                                hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);   // pageable source: copy must finish before `host` dies
  if (e == hipSuccess) {
    launch_compose(stream, (int)P, M, dev_tab, num_segments, means3D, scales, rotations, opacities, shs);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  (void)hipFree(dev_tab);
  if (e != hipSuccess) return fail(GRPG_ERR_HIP, std::string("grpg_compose: ") + hipGetErrorString(e));
  return GRPG_OK;
}

int grpg_backward(int P, int D, int M, int R, int S, const float* background, int width,
                  int height, const float* means3D, const float* shs, const float* colors_precomp,
                  const float* semantics, const float* alphas, const float* scales,
                  float scale_modifier, const float* rotations, const float* cov3D_precomp,
                  const float* viewmatrix, const float* projmatrix, const float* campos,
                  float tan_fovx, float tan_fovy, const int* radii, char* geom_buffer,
                  char* binning_buffer, char* image_buffer, const float* dL_dpix,
                  const float* dL_dpix_depth, const float* dL_dalphas,
                  const float* dL_dpix_semantic, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D,
                  float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                  float* dL_dsemantic, int debug, void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P <= 0) return GRPG_OK;
  if (!geom_buffer || !binning_buffer || !image_buffer)
    return fail(GRPG_ERR_BAD_BUFFER, "NULL state buffer");
  if (int rc = geom_check_backward(geom_buffer, P, (hipStream_t)hip_stream)) return rc;
  for (auto& hw : g_host_words)
    if (int rc = check_async_error(&hw)) return rc;
  // dL_dconic / dL_ddepth (pure intermediates of the reference's binding) and dL_dcolor / dL_dcov3D
  // (gradients of the OPTIONAL inputs colors_precomp / cov3D_precomp) may be NULL: not written then
  if (!dL_dpix || !dL_dpix_depth || !dL_dalphas || !alphas || !dL_dmean2D || !dL_dopacity || !dL_dmean3D)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL gradient pointer");
  if ((colors_precomp && !dL_dcolor) || (cov3D_precomp && !dL_dcov3D))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "colors_precomp / cov3D_precomp given without their gradient array");
  // every gradient array is WRITTEN, not accumulated: the inputs in use need theirs
  if (!colors_precomp && (!shs || !dL_dsh))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "shs (no colors_precomp) needs shs and dL_dsh");
  if (!cov3D_precomp && (!scales || !rotations || !dL_dscale || !dL_drot))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "scales / rotations (no cov3D_precomp) need dL_dscale and dL_drot");
  if (!means3D || !background || !viewmatrix || !projmatrix || !campos)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL means3D / camera / background pointer");
  if (S > 0 && (!semantics || !dL_dpix_semantic || !dL_dsemantic))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL semantic pointer with S>0");
  if (S > GRPG_MAX_SEMANTIC_BACKWARD)   // the blend backward carries the channels in registers
    return fail(GRPG_ERR_INVALID_ARGUMENT,
                "backward supports at most 32 semantic channels (the reference: 20, config.h:16)");
  hipStream_t stream = (hipStream_t)hip_stream;
  const CameraArgs cam = make_camera(viewmatrix, projmatrix, campos, width, height, tan_fovx, tan_fovy);
  const uint32_t T = (uint32_t)cam.gx * (uint32_t)cam.gy;
  const GeomLayout GL = geom_layout((size_t)P);
  const ImgLayout IL = img_layout(T, (size_t)width * height);
  if (debug) {  // validate the blobs (costs a sync; only in debug mode, like the reference's checks)
    BlobHeader h[3];
    HIP_TRY(hipMemcpyAsync(&h[0], geom_buffer, sizeof(BlobHeader), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&h[1], binning_buffer, sizeof(BlobHeader), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&h[2], image_buffer, sizeof(BlobHeader), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (h[0].magic != GEOM_MAGIC || h[1].magic != BIN_MAGIC || h[2].magic != IMG_MAGIC ||
        h[0].P != (uint32_t)P || h[0].R != (uint32_t)R || h[1].Rcap < (uint32_t)R ||
        h[2].W != (uint32_t)width || h[2].H != (uint32_t)height)
      return fail(GRPG_ERR_BAD_BUFFER, "state buffers do not match this call (P/R/W/H or magic)");
    if (h[0].has_grad_rec == 0u)
      return fail(GRPG_ERR_BAD_BUFFER, "the geometry buffer was carved by an evaluation forward (no gradient records)");
    if (h[2].pc_timeout != 0u)
      return fail(GRPG_ERR_HIP, "the forward of this frame reported a producer/consumer timeout: its image is invalid");
  }
  const RecView rec = {(const float4*)(geom_buffer + GL.rec)};
  const int* radii_int = radii ? radii : (const int*)(geom_buffer + GL.radii);
  // the point list is the first array of the binning blob whatever capacity it was carved for
  const uint32_t* point_list = (const uint32_t*)(binning_buffer + bin_layout(0).val_a);
  const uint2* ranges = (const uint2*)(image_buffer + IL.ranges);
  const uint32_t* n_contrib = (const uint32_t*)(image_buffer + IL.n_contrib);

  // per-Gaussian gradient records (one 64-byte line each) in the tail of the geometry blob: cleared,
  // accumulated by the blend backward, fanned out into the caller's arrays by the preprocess backward
  float* grad_rec = (float*)(geom_buffer + GL.grad_rec);
  HIP_TRY(hipMemsetAsync(grad_rec, 0, (size_t)P * GRAD_STRIDE * sizeof(float), stream));
  BwdTimingRecord* const btr = bwd_timing_begin(stream);
  BwdTimingGuard btg(btr);
  launch_render_backward(stream, ranges, point_list, rec, semantics, S, width, height, cam.gx,
                         cam.gy, background, alphas, n_contrib,
                         (const uint32_t*)(image_buffer + IL.work), dL_dpix, dL_dpix_depth, dL_dalphas,
                         dL_dpix_semantic, grad_rec, dL_dsemantic, (const BlobHeader*)binning_buffer,
                         (const uint32_t*)(image_buffer + IL.ck_count),
                         (const uint32_t*)(image_buffer + IL.bwd_ctl), (uint32_t)R);
  STAGE_CHECK("render backward");
  bwd_timing_mark(btr, 1, stream);
  launch_preprocess_backward(stream, P, D, M, means3D, radii_int, colors_precomp ? nullptr : shs,
                             rec, cov3D_precomp ? nullptr : scales, rotations, scale_modifier,
                             cov3D_precomp, cam, grad_rec, dL_dmean2D, dL_dconic, dL_dopacity,
                             dL_dmean3D, dL_dcolor, dL_ddepth, dL_dcov3D,
                             colors_precomp ? nullptr : dL_dsh, dL_dscale, dL_drot);
  bwd_timing_mark(btr, 2, stream);
  btg.done = true;
  STAGE_CHECK("preprocess backward");
  return GRPG_OK;
}

int grpg_mark_visible(int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, unsigned char* present, void* hip_stream) {
  (void)projmatrix;
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present)))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "bad argument");
  launch_mark_visible((hipStream_t)hip_stream, P, means3D, viewmatrix, present);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

int grpg_visible_filter(int P, int M, int width, int height, const float* means3D,
                        const float* scales, float scale_modifier, const float* rotations,
                        const float* cov3D_precomp, const float* viewmatrix,
                        const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered,
                        int* radii, float* means2D, int debug, void* hip_stream) {
  (void)M; (void)prefiltered;
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P < 0 || width <= 0 || height <= 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "bad size");
  if (P == 0) return GRPG_OK;
  if (!means3D || !viewmatrix || !projmatrix || !radii || !means2D ||
      (!cov3D_precomp && (!scales || !rotations)))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t stream = (hipStream_t)hip_stream;
  const CameraArgs cam = make_camera(viewmatrix, projmatrix, nullptr, width, height, tan_fovx, tan_fovy);
  launch_visible_filter(stream, P, means3D, scales, scale_modifier, rotations, cov3D_precomp, cam,
                        radii, means2D);
  STAGE_CHECK("visible filter");
  return GRPG_OK;
}

int grpg_pack_rgb_u8(const float* src, unsigned char* dst, size_t n, void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (n > 0 && (!src || !dst)) return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 3))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "src must be 16-byte and dst 4-byte aligned");
  launch_pack_u8((hipStream_t)hip_stream, src, dst, n);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

int grpg_pack_rgb_u8_hwc(const float* src_chw, unsigned char* dst_hwc, int height, int width,
                         int truncate, void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (height < 0 || width < 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "negative size");
  const size_t npix = (size_t)height * (size_t)width;
  if (npix > 0 && (!src_chw || !dst_hwc)) return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (((uintptr_t)src_chw & 15) || ((uintptr_t)dst_hwc & 3))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "src must be 16-byte and dst 4-byte aligned");
  launch_pack_hwc((hipStream_t)hip_stream, src_chw, dst_hwc, npix, truncate);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

int grpg_sky_composite_ex(const float* cube, int res, const float* ray_matrix, int ray_matrix_on_device,
                          float fill, int clamp_out, int width, int height, const float* rgb_in,
                          const float* acc, const unsigned char* mask, const float* jitter,
                          float* rgb_out, float* sky_out, void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (!cube || !ray_matrix || res <= 0 || width <= 0 || height <= 0)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "bad cube / ray matrix / size");
  if ((rgb_in == nullptr) != (rgb_out == nullptr))
    return fail(GRPG_ERR_INVALID_ARGUMENT, "rgb_in and rgb_out go together");
  if (!rgb_out && !sky_out) return fail(GRPG_ERR_INVALID_ARGUMENT, "nothing to write");
  launch_sky_composite((hipStream_t)hip_stream, cube, res, ray_matrix, ray_matrix_on_device, fill,
                       clamp_out, width, height, rgb_in, acc, mask, jitter, rgb_out, sky_out);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

int grpg_sky_composite(const float* cube, int res, const float* ray_matrix, float fill,
                       int clamp_out, int width, int height, const float* rgb_in, const float* acc,
                       float* rgb_out, float* sky_out, void* hip_stream) {
  return grpg_sky_composite_ex(cube, res, ray_matrix, 0, fill, clamp_out, width, height, rgb_in, acc,
                               nullptr, nullptr, rgb_out, sky_out, hip_stream);
}

int grpg_sky_backward_ex(const float* cube, int res, const float* ray_matrix, int ray_matrix_on_device,
                         float fill, int width, int height, const float* acc, const unsigned char* mask,
                         const float* jitter, const float* grad_rgb, float* grad_cube, float* grad_acc,
                         void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (!cube || !ray_matrix || !grad_rgb || res <= 0 || width <= 0 || height <= 0)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "bad cube / ray matrix / gradient / size");
  launch_sky_backward((hipStream_t)hip_stream, cube, res, ray_matrix, ray_matrix_on_device, fill, width,
                      height, acc, mask, jitter, grad_rgb, grad_cube, grad_acc);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

int grpg_sky_backward(const float* cube, int res, const float* ray_matrix, float fill, int width,
                      int height, const float* acc, const float* grad_rgb, float* grad_cube,
                      float* grad_acc, void* hip_stream) {
  return grpg_sky_backward_ex(cube, res, ray_matrix, 0, fill, width, height, acc, nullptr, nullptr,
                              grad_rgb, grad_cube, grad_acc, hip_stream);
}

size_t grpg_knn_workspace_bytes(int P) { return knn_workspace_bytes(P); }

int grpg_knn_mean_dist2(int P, const float* points, float* mean_dists,
                        grpg_alloc_fn workspace_alloc, void* workspace_user, void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P < 0) return fail(GRPG_ERR_INVALID_ARGUMENT, "negative size");
  if (P == 0) return GRPG_OK;
  if (!points || !mean_dists) return fail(GRPG_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (!workspace_alloc) return fail(GRPG_ERR_INVALID_ARGUMENT, "workspace allocator must not be NULL");
  char* ws = workspace_alloc(knn_workspace_bytes(P), workspace_user);
  if (!ws) return fail(GRPG_ERR_ALLOC, "knn workspace allocation failed");
  if ((uintptr_t)ws & 255) return fail(GRPG_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
  launch_knn((hipStream_t)hip_stream, P, points, mean_dists, ws);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

int grpg_debug_export(int P, int R, int width, int height, const char* geom_buffer,
                      const char* binning_buffer, const char* image_buffer, uint64_t* keys_sorted,
                      uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib, float* means2D,
                      float* depths, float* conic_opacity, float* rgb, uint32_t* tiles_touched,
                      void* hip_stream) {
  g_last_error.clear();
  if (int rc = ensure_device()) return rc;
  if (P < 0 || R < 0 || !geom_buffer || !binning_buffer || !image_buffer)
    return fail(GRPG_ERR_INVALID_ARGUMENT, "bad argument");
  hipStream_t stream = (hipStream_t)hip_stream;
  const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
  const GeomLayout GL = geom_layout((size_t)P);
  // the binning blob was carved for a capacity >= R: read it from the blob's header
  BlobHeader bh;
  HIP_TRY(hipMemcpyAsync(&bh, binning_buffer, sizeof(BlobHeader), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (bh.magic != BIN_MAGIC || bh.Rcap < (uint32_t)R)
    return fail(GRPG_ERR_BAD_BUFFER, "binning buffer does not match this call");
  // hierarchical blob: the header's Rc is the coarse capacity it was carved for
  const BinLayout BL = bh.hier ? bin_layout((size_t)bh.Rcap, (size_t)gx * gy, super_tiles(gx, gy), (size_t)bh.Rc)
                               : bin_layout((size_t)bh.Rcap);
  const ImgLayout IL = img_layout((size_t)gx * gy, (size_t)width * height);
  launch_debug_export(stream, P, (uint32_t)R, width, height, gx, gy,
                      RecView{(const float4*)(geom_buffer + GL.rec)},
                      (const uint32_t*)(geom_buffer + GL.tiles),
                      (const uint32_t*)(binning_buffer + BL.key_a),
                      bh.hier ? (const uint32_t*)(binning_buffer + BL.tile_start) : nullptr,
                      (const uint32_t*)(binning_buffer + BL.val_a),
                      (const uint2*)(image_buffer + IL.ranges),
                      (const uint32_t*)(image_buffer + IL.n_contrib), keys_sorted, point_list,
                      ranges, n_contrib, means2D, depths, conic_opacity, rgb, tiles_touched);
  HIP_TRY(hipGetLastError());
  return GRPG_OK;
}

}  // extern "C"
