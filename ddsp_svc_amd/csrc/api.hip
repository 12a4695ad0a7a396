// C ABI of libddsp_hip.so (declared in include/ddsp_hip.h): argument checks, workspace carving and
// the launch sequences of the two synthesiser tails.  No allocation, no synchronisation.
#include "../../include/ddsp_hip.h"
#include "kernels.h"
#include "philox.h"
#include <atomic>
#include <stdio.h>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <string.h>

using namespace ddsp;

// ---- tuning knobs (csrc/tuning.h): environment read once, afterwards only ddsp_hip_set_tuning() ----------------------
namespace ddsp {
thread_local int t_geometry_batch = 0;
namespace {
const char* const kKnobNames[KNOB_COUNT] = {"BLK_WPS", "BLK_RUN", "BLK_PADLDS", "FFT_RUN", "STFT_WPS", "STFT_RUN",
                                            "MEL_WPS", "MEL_RUN", "FIR_MAX_SLOTS", "SINS_V1", "TAPS_GEMM", "STREAM_LAYOUT",
                                            "BLK_TURNS", "CZT_ROUNDS", "CZT_TURNS", "SINS_NOSKIP", "SMALL_PATH", "LANE_ROWS", "LANES", "FIR_BWD_DIRECT", "BWD_WPS", "TAPS_FULL",
                                            "AP_BWD_SPLIT", "SINS_SEQ"};
std::atomic<long> g_knobs[KNOB_COUNT];
std::once_flag g_knobs_once;
// A knob whose kernel generation is not compiled into this build (the product library ships ONE generation per kernel; the
// superseded ones exist only under -DDDSP_AB_GENERATIONS, tools/build_variant.sh) does nothing: setting it is an ERROR from
// ddsp_hip_set_tuning and a warning from the environment, so that an A/B run cannot measure the same kernel twice under two names.
bool knob_is_inert(int i, long v) {
#ifdef DDSP_AB_GENERATIONS
  (void)i; (void)v;
  return false;
#else
  if (v == 0) return false;
  return i == KNOB_BLK_WPS || i == KNOB_BLK_PADLDS || (i == KNOB_SINS_V1 && v == 2);
#endif
}
void knobs_from_env() {
  for (int i = 0; i < KNOB_COUNT; ++i) {
    char name[64] = "DDSP_HIP_";
    strncat(name, kKnobNames[i], sizeof(name) - strlen(name) - 1);
    const char* e = getenv(name);
    long v = e ? atol(e) : 0;
    if (knob_is_inert(i, v)) {
      fprintf(stderr, "libddsp_hip: %s=%ld ignored: that kernel generation is not in this build (-DDDSP_AB_GENERATIONS)\n", name, v);
      v = 0;
    }
    g_knobs[i].store(v, std::memory_order_relaxed);
  }
}
int knob_index(const char* name) {
  if (!name) return -1;
  if (strncmp(name, "DDSP_HIP_", 9) == 0) name += 9;
  for (int i = 0; i < KNOB_COUNT; ++i)
    if (strcmp(name, kKnobNames[i]) == 0) return i;
  return -1;
}
}  // namespace
long knob(Knob k) {
  std::call_once(g_knobs_once, knobs_from_env);
  return g_knobs[k].load(std::memory_order_relaxed);
}
int knob_set(const char* name, long v) {
  std::call_once(g_knobs_once, knobs_from_env);
  const int i = knob_index(name);
  if (i < 0 || knob_is_inert(i, v)) return -1;
  g_knobs[i].store(v, std::memory_order_relaxed);
  return 0;
}
long knob_get(const char* name) {
  const int i = knob_index(name);
  return i < 0 ? -1 : knob((Knob)i);
}
}  // namespace ddsp

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int finish() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Carver {
  char* base;
  size_t size, used;
  bool ok;
  Carver(void* p, size_t n) : base(static_cast<char*>(p)), size(n), used(0), ok(p != nullptr || n == 0) {}
  template <class T> T* take(size_t count) {
    size_t off = align_up(used, 256);
    size_t end = off + count * sizeof(T);
    if (end > size) { ok = false; used = end; return nullptr; }
    used = end;
    return reinterpret_cast<T*>(base + off);
  }
};

struct SynthWs {
  float *buf0, *buf1, *taps, *re, *im;
  float *taps_nz, *nzbuf;      // the noise branch's own taps and output when it runs on a second stream
  float *taps3;                // streaming shapes only (B F < kSmallRows): a third tap buffer, so that all tap syntheses of a step are one launch
};

// Fork / join of independent branches of a synthesiser tail onto a caller-provided second stream.  The events are
// created once per host thread and device (the only thing this library ever creates) and re-used: a wait captures the
// record that precedes it, so re-recording an event for the next call does not disturb waits already enqueued.
struct BranchEvents {
  hipEvent_t fork = nullptr, join = nullptr, mid[2] = {nullptr, nullptr};
};

// The HIP objects this library creates (fork / join events, the second lane's streams) live in ONE process-wide pool per device:
// a call checks a set out and hands it back when it returns, so their number is bounded by the calls in flight at once -- not by
// the host threads that ever called (rounds 2 - 5 kept them in thread_local arrays without destructors: a caller on short-lived
// threads leaked a set per thread).  A set that goes back while its work is still in flight is safe to re-use: a wait captures
// the record that precedes it, and streams order whatever the next user enqueues behind it.  Nothing is destroyed at exit.
template <class T>
struct DevicePool {
  std::mutex m;
  std::vector<T*> idle[64];
  T* take(int dev) {
    std::lock_guard<std::mutex> lock(m);
    if (idle[dev].empty()) return nullptr;
    T* t = idle[dev].back();
    idle[dev].pop_back();
    return t;
  }
  void give(int dev, T* t) {
    std::lock_guard<std::mutex> lock(m);
    idle[dev].push_back(t);
  }
};
DevicePool<BranchEvents> g_branch_pool;

struct Branch {
  hipStream_t main, aux;
  BranchEvents* ev = nullptr;
  int dev = -1;
  bool forked = false;
  Branch(hipStream_t m, void* aux_stream) : main(m), aux(m) {
    if (!aux_stream || S(aux_stream) == m) return;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return;
    BranchEvents* e = g_branch_pool.take(d);
    if (!e) {
      hipEvent_t made[4] = {nullptr, nullptr, nullptr, nullptr};
      for (int i = 0; i < 4; ++i)
        if (hipEventCreateWithFlags(&made[i], hipEventDisableTiming) != hipSuccess) {
          for (int j = 0; j < i; ++j) (void)hipEventDestroy(made[j]);
          return;
        }
      e = new BranchEvents;
      e->fork = made[0]; e->join = made[1]; e->mid[0] = made[2]; e->mid[1] = made[3];
    }
    ev = e;
    dev = d;
    // everything enqueued on the main stream so far (the inputs' producers) precedes the branch
    if (hipEventRecord(e->fork, m) != hipSuccess) return;
    if (hipStreamWaitEvent(S(aux_stream), e->fork, 0) != hipSuccess) return;
    aux = S(aux_stream);
    forked = true;
  }
  ~Branch() {
    if (ev) g_branch_pool.give(dev, ev);
  }
  Branch(const Branch&) = delete;
  Branch& operator=(const Branch&) = delete;
  // a point of the branch the main stream waits for while the branch keeps running (i = 0, 1)
  void publish(int i) {
    if (forked) (void)hipEventRecord(ev->mid[i], aux);
  }
  void await(int i) {
    if (forked) (void)hipStreamWaitEvent(main, ev->mid[i], 0);
  }
  // the main stream continues only after everything the branch enqueued
  void join() {
    if (!forked) return;
    (void)hipEventRecord(ev->join, aux);
    (void)hipStreamWaitEvent(main, ev->join, 0);
    forked = false;
  }
};

// ---- Sub-batches and lanes (round 5; OPT-IN: knob LANE_ROWS) -------------------------------------------------------------
// Utterances are independent (core.py:120-182 has no op across the batch), so a large call CAN be issued as sub-batches of
// ~LANE_ROWS frames that alternate between two LANES: lane 0 = the caller's stream pair, lane 1 = a pair of streams this
// library owns (created once per host thread and device, beside the fork / join events).  Each lane has ONE workspace slot
// that its sub-batches re-use in stream order, so the scratch of a call is two slots of a sub-batch whatever B is (B = 64 x
// 10 s with LANE_ROWS = 14336: the bytes of B = 32) and the prefix of sub-batch k+1 (exciter, tap syntheses) runs beside the
// filters of sub-batch k.  The samples are those of the unsplit call bit for bit (tests/test_lanes.py).
// MEASURED (same box, profiles/r05_v1_*): it LOSES at every batch size -- B = 32: 0.351 ms against 0.317 unsplit (two lanes of
// 16), 0.371 in sequence on one lane; B = 64: 0.676 (four of 16) / 0.646 (two of 32) against 0.617; B = 256: 2.62 against 2.28.
// Every kernel of the step already fills the chip, and two of them side by side take as long as one after the other plus the
// contention (filter launches 135 - 159 us beside each other against 84 alone); what a step pays per launch (ramp, drain,
// dependency latency) is paid per sub-batch.  Unsplit, the step's rate RISES with B (3.76e10 samples/s at 16 utterances,
// 4.46e10 at 32, 4.59e10 at 64, 4.95e10 at 256): t = 36 us + 8.77 us per utterance.  So the default is ONE batch; the split
// stays for callers that must bound the scratch (0.34 GB per 32 utterances of 10 s).
// Knobs: LANE_ROWS = frames per sub-batch (0 / 1: never split), LANES = 1: one lane (sub-batches in sequence, one slot).

struct LanePlan { int nsub, Bs, slots; };

LanePlan lane_plan(int B, int F) {
  LanePlan p{1, B, 1};
  const long target = knob(KNOB_LANE_ROWS);
  const long R = (long)B * F;
  if (target <= 1 || B < 2) return p;
  if (2 * R < 3 * target) return p;                       // below one and a half sub-batches: not worth two half-empty lanes
  long nsub = (R + target - 1) / target;                  // sub-batches of at most `target` frames, evenly sized
  if (nsub > B) nsub = B;
  p.Bs = (int)((B + nsub - 1) / nsub);
  // the tap syntheses transform two ROWS per complex transform (ir_pfa.hip, ir_czt.hip), and a row's last bits depend on its
  // partner: a sub-batch starts on an even row, so that every row keeps the partner it has in the unsplit call
  if ((F & 1) && (p.Bs & 1)) ++p.Bs;
  p.nsub = (B + p.Bs - 1) / p.Bs;
  if (p.nsub < 2) return LanePlan{1, B, 1};
  p.slots = knob(KNOB_LANES) == 1 ? 1 : 2;
  if (p.slots > p.nsub) p.slots = p.nsub;
  return p;
}

struct LaneSet {
  hipStream_t main1 = nullptr, aux1 = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};

// a second lane's stream pair on the current device, checked out of the pool (give it back with release_lane), or null (then
// every sub-batch takes lane 0)
DevicePool<LaneSet> g_lane_pool;
void release_lane(LaneSet* l, int dev) {
  if (l) g_lane_pool.give(dev, l);
}
LaneSet* second_lane(int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  *dev_out = dev;
  if (LaneSet* have = g_lane_pool.take(dev)) return have;
  LaneSet& l = *new LaneSet;
  if (!l.join) {
    hipStream_t s0 = nullptr, s1 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool ok = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking) == hipSuccess &&
                    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      if (s0) (void)hipStreamDestroy(s0);
      if (s1) (void)hipStreamDestroy(s1);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      (void)hipGetLastError();
      delete &l;
      return nullptr;
    }
    l.main1 = s0; l.aux1 = s1; l.fork = e0; l.join = e1;
  }
  return &l;
}

// Whether this call uses the dense contraction at 256 bins too (knob TAPS_GEMM): read ONCE per entry point into this
// thread-local, so that the stream layout and every tap synthesis of one call agree even if another thread changes
// the knob meanwhile (a disagreement would let the fallback stage the all-pass response in a buffer the layout has
// already given to the exciter).
thread_local bool t_taps_gemm = false;
thread_local int t_czt_from = 112;          // chirp-z form from this bin count on (below it the dense contraction is the faster one:
                                            // its cost falls with n^2, the 512-point plan's does not; equal at ~120 bins, measured)
struct TapsFormScope {
  TapsFormScope() {
    const long v = knob(KNOB_TAPS_GEMM);     // 0: by shape; 1: the dense contraction everywhere; 2: chirp-z wherever its plans reach
    t_taps_gemm = v == 1;
    t_czt_from = v == 2 ? 2 : 112;
  }
};

// the periodic Hann behind the two basis planes of the table (k_ir_table, ir.hip)
const float* table_hann(const float* table, int n) {
  const long KP = ((long)n + 15) / 16 * 16, NP = ((long)n + 255) / 256 * 256;
  return table + 2 * KP * NP;
}

// tap synthesis of one filter: the prime-factor form when the shape is its (n_mag = 256), the chirp-z form for the other bin
// counts from 112 to 1025, else (or under knob TAPS_GEMM = 1) the dense contraction
void synth_taps(const float* a_re, long ld_re, const float* a_im, long ld_im, int act, float scale, const float* table,
                int mode, const float* half_width, long rows, int n, float* taps, hipStream_t st, float hw_sr = 0.f) {
  if (!t_taps_gemm &&
      launch_taps_pfa510(a_re, ld_re, a_im, ld_im, 0, act, scale, table, mode, half_width, rows, n, taps, st, hw_sr) == 0)
    return;
  if (!t_taps_gemm && n != 256 && n >= t_czt_from &&
      launch_taps_czt(a_re, ld_re, a_im, ld_im, act, scale, table_hann(table, n), mode, half_width, rows, n, taps, st, hw_sr) == 0)
    return;
  launch_ir_gemm(a_re, ld_re, a_im, ld_im, act, scale, table, mode, half_width, rows, n, taps, st, hw_sr);
}

// all-pass taps from the raw group-delay control (vocoder.py:581,599 / :834,845): fused in the prime-factor kernel, else
// response (re, im scratch of rows * n floats each) + dense contraction
void synth_allpass_taps(const float* c_gd, long ld_gd, const float* table, long rows, int n, float* re, float* im,
                        float* taps, hipStream_t st) {
  if (!t_taps_gemm && launch_taps_pfa510(c_gd, ld_gd, nullptr, 0, 1, DDSP_HIP_ACT_NONE, 1.0f, table, DDSP_HIP_MODE_ROLL, nullptr,
                                         rows, n, taps, st) == 0)
    return;
  launch_allpass_response(c_gd, ld_gd, rows, n, re, im, st);
  if (!t_taps_gemm && n != 256 && n >= t_czt_from &&
      launch_taps_czt(re, n, im, n, DDSP_HIP_ACT_NONE, 1.0f, table_hann(table, n), DDSP_HIP_MODE_ROLL, nullptr, rows, n, taps, st, 0.f) == 0)
    return;
  launch_ir_gemm(re, n, im, n, DDSP_HIP_ACT_NONE, 1.0f, table, DDSP_HIP_MODE_ROLL, nullptr, rows, n, taps, st);
}

// knob STREAM_LAYOUT = 1 / 4: batch shapes keep the two-stream layouts of rounds 2 - 5 (A/B runs); default: the fused layout
bool fused_off() {
  const long v = knob(KNOB_STREAM_LAYOUT);
  return v == 1 || v == 4;
}

// does a CombSub / Sins call of this shape take the fused one-stream layout (combsub_rows / sins_rows)?  One predicate for the
// launch path and for ddsp_hip_tail_layout, which tells a training caller where the call left its intermediates.
bool fused_shape_ok(long R, int F, int hop, int n0, int n1, int n2, int fir_impl, bool gen_on, bool combsub) {
  if (!(R < kSmallRows || !fused_off()) || hop != 512 || t_taps_gemm || knob(KNOB_SMALL_PATH) == 1) return false;
  if (combsub) {
    (void)gen_on;                                            // (the in-kernel noise draw rides in the paired filter launch as its second job)
    if (n0 != 256 || n1 != 256 || n2 != 256 || !(fir_impl == 0 || fir_impl == 5)) return false;
    if ((long)F * hop > (1L << 24) || R >= (1L << 31) - 64) return false;       // the exciter job's shift form (make_exciter_job)
#ifdef DDSP_AB_GENERATIONS                                   // (the two-wave kernel of the A/B builds takes no second job: launch_fir_blk's predicate)
    if (!((knob(KNOB_BLK_WPS) == 0 || knob(KNOB_BLK_WPS) >= 3) && knob(KNOB_BLK_PADLDS) == 0)) return false;
#endif
    return true;
  }
  return n1 == 256 && n2 == 256;                              // Sins: all-pass and noise filter at 256 bins (n0 = harmonics)
}

// The fused layouts keep the tap rows of the NOISE filter (zero phase, Hann window: an even response, tap N - j is tap j) as the first
// n = N/2 + 1 taps, and the filter reads them so (launch_taps_pfa510's half_rows, launch_fir_blk's taps_half) -- 56 MB less through
// HBM per step at B = 32 x 10 s and half the window factors and stores of that tap synthesis.  (CombSub's harmonic filter is zero
// phase too, but its dynamic window clamps the upper side only, core.py:245: not even.)  Knob TAPS_FULL = 1: whole rows (same-box
// A/Bs; same bits either way: the whole rows hold the same numbers twice).  No backward kernel reads these taps.
bool taps_half_ok(int F, int hop, int fir_impl) {
#if defined(DDSP_AB_GENERATIONS) || defined(DDSP_PFA_NOSYM)
  (void)F; (void)hop; (void)fir_impl;
  return false;
#else
  return knob(KNOB_TAPS_FULL) != 1 && hop == 512 && (fir_impl == 0 || fir_impl == 5) && (long)F * hop < (1L << 28);
#endif
}

size_t carve_synth(Carver& c, int B, int F, int hop, int n_max, SynthWs& w) {
  const size_t BT = (size_t)B * F * hop, R = (size_t)B * F, N = 2 * (size_t)(n_max - 1);
  w.buf0 = c.take<float>(BT);
  w.buf1 = c.take<float>(BT);
  w.taps = c.take<float>(R * N);
  // The all-pass response (re, im) is dead once its taps are synthesised, before the exciter is written: when it
  // fits (2 n <= hop) it lives in buf0, and the harmonic signal re-uses buf0 after the first filter has consumed
  // the exciter -- the step then cycles through three [B,T]-sized buffers instead of six (less of it falls out of
  // the 256 MB MALL between kernels).
  if (2 * R * n_max <= BT) {
    w.re = w.buf0;
    w.im = w.buf0 + R * n_max;
  } else {
    w.re = c.take<float>(R * n_max);
    w.im = c.take<float>(R * n_max);
  }
  w.taps_nz = c.take<float>(R * N);
  w.nzbuf = c.take<float>(BT);
  // a third tap buffer: all tap syntheses of a step are then ONE launch (the fused layout of ddsp_hip_combsub_synth: 256-bin models
  // at hop 512; knob STREAM_LAYOUT 1 / 4 = the two-stream layouts of rounds 2 - 5, which do not need it)
  w.taps3 = ((long)R < kSmallRows || (n_max == 256 && hop == 512 && !fused_off())) ? c.take<float>(R * N) : nullptr;
  return align_up(c.used, 256);
}

}  // namespace

extern "C" {

int ddsp_hip_version(void) { return DDSP_HIP_VERSION; }

const char* ddsp_hip_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case DDSP_HIP_EINVAL: return "ddsp_hip: invalid argument (null pointer or non-positive size)";
    case DDSP_HIP_EHOP: return "ddsp_hip: hop > 2048 is not supported by the phase scan";
    case DDSP_HIP_ESHAPE: return "ddsp_hip: shape not supported by the requested kernel";
    case DDSP_HIP_EWS: return "ddsp_hip: workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "ddsp_hip: unknown error";
  }
}

int ddsp_hip_set_tuning(const char* name, long value) { return knob_set(name, value) == 0 ? 0 : DDSP_HIP_EINVAL; }

long ddsp_hip_get_tuning(const char* name) { return knob_get(name); }

int ddsp_hip_upsample(const float* sig, int B, int F, int C, int hop, float* out, void* stream) {
  if (B < 0 || F <= 0 || C <= 0 || hop <= 0) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!sig || !out) return DDSP_HIP_EINVAL;
  launch_upsample(sig, B, F, C, hop, out, S(stream));
  return finish();
}

int ddsp_hip_remove_above_fmax(const float* amps, const float* pitch, long rows, int H, float fmax, int level_start,
                               float* out, void* stream) {
  if (rows < 0 || H <= 0) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!amps || !pitch || !out) return DDSP_HIP_EINVAL;
  launch_remove_above_fmax(amps, pitch, rows, H, fmax, level_start, out, S(stream));
  return finish();
}

int ddsp_hip_phase(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                   double* frame_sums, double* phase0, float* phase_frames, float* x_or_null, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !frame_sums || !phase0) return DDSP_HIP_EINVAL;
  if (launch_phase(f0_frames, initial_phase, B, F, hop, sr, infer, frame_sums, phase0, phase_frames, x_or_null,
                   S(stream)) != 0)
    return DDSP_HIP_EHOP;
  return finish();
}

size_t ddsp_hip_ir_table_bytes(int n_mag) { return n_mag >= 2 ? ir_table_floats(n_mag) * sizeof(float) : 0; }

int ddsp_hip_ir_table(int n_mag, float* table, void* stream) {
  if (n_mag < 2 || !table) return DDSP_HIP_EINVAL;
  launch_ir_table(n_mag, table, S(stream));
  return finish();
}

int ddsp_hip_allpass_response(const float* c, long ld, long rows, int n_mag, float* re, float* im, void* stream) {
  if (rows < 0 || n_mag < 2 || ld < n_mag) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!c || !re || !im) return DDSP_HIP_EINVAL;
  launch_allpass_response(c, ld, rows, n_mag, re, im, S(stream));
  return finish();
}

int ddsp_hip_impulse_response(const float* resp_re, long ld_re, const float* resp_im, long ld_im, int act, float scale,
                              int mode, const float* half_width, long rows, int n_mag, const float* table, float* taps,
                              void* stream) {
  if (rows < 0 || n_mag < 2 || ld_re < n_mag || (resp_im && ld_im < n_mag)) return DDSP_HIP_EINVAL;
  if (mode < 0 || mode > 2 || act < 0 || act > 1) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!resp_re || !table || !taps) return DDSP_HIP_EINVAL;
  if (mode == DDSP_HIP_MODE_DYNAMIC && !half_width) return DDSP_HIP_EINVAL;
  const TapsFormScope form;
  synth_taps(resp_re, ld_re, resp_im, ld_im, act, scale, table, mode, half_width, rows, n_mag, taps, S(stream));
  return finish();
}

size_t ddsp_hip_allpass_taps_scratch_bytes(long rows, int n_mag) {
  if (rows <= 0 || n_mag < 2) return 0;
  return align_up((size_t)rows * n_mag * sizeof(float), 256) * 2;
}

int ddsp_hip_allpass_taps(const float* c, long ld, long rows, int n_mag, const float* table, float* taps, void* scratch,
                          size_t scratch_bytes, void* stream) {
  if (rows < 0 || n_mag < 2 || ld < n_mag) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!c || !table || !taps || !scratch) return DDSP_HIP_EINVAL;
  if (scratch_bytes < ddsp_hip_allpass_taps_scratch_bytes(rows, n_mag)) return DDSP_HIP_EWS;
  float* re = static_cast<float*>(scratch);
  float* im = reinterpret_cast<float*>(static_cast<char*>(scratch) + align_up((size_t)rows * n_mag * sizeof(float), 256));
  const TapsFormScope form;
  synth_allpass_taps(c, ld, table, rows, n_mag, re, im, taps, S(stream));
  return finish();
}

int ddsp_hip_impulse_response_backward(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, int mode,
                                       const float* half_width, long rows, int n_mag, const float* table, float* d_re,
                                       float* d_im, void* stream) {
  if (rows < 0 || n_mag < 2 || mode < 0 || mode > 2 || act < 0 || act > 1) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!d_taps || !table || !d_re) return DDSP_HIP_EINVAL;
  if (act == DDSP_HIP_ACT_EXP && (!ctrl || ld_ctrl < n_mag)) return DDSP_HIP_EINVAL;
  if (mode == DDSP_HIP_MODE_DYNAMIC && !half_width) return DDSP_HIP_EINVAL;
  if (d_im && act != DDSP_HIP_ACT_NONE) return DDSP_HIP_ESHAPE;
  const TapsFormScope form;
  // the prime-factor form at 256 bins, the chirp-z form at the other bin counts up to 1025, else the dense contraction
  const bool fast = !t_taps_gemm &&
      (launch_taps_pfa510_bwd(d_taps, ctrl, ld_ctrl, act, scale, table, mode, half_width, rows, n_mag, d_im != nullptr, d_re, d_im,
                              S(stream)) == 0 ||
       (n_mag != 256 && n_mag >= t_czt_from && launch_taps_czt_bwd(d_taps, ctrl, ld_ctrl, act, scale, table_hann(table, n_mag), mode, half_width, rows,
                                            n_mag, d_re, d_im, S(stream)) == 0));
  if (!fast)
    launch_ir_gemm_bwd(d_taps, ctrl, ld_ctrl, act, scale, table, mode, half_width, rows, n_mag, d_im != nullptr, d_re, d_im,
                       S(stream));
  return finish();
}

int ddsp_hip_window_impulse_response(const float* ir, int mode, const float* half_width, long rows, int N, float* out,
                                     void* stream) {
  if (rows < 0 || N < 1 || mode < 0 || mode > 2) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!ir || !out || ir == out) return DDSP_HIP_EINVAL;
  if (mode == DDSP_HIP_MODE_DYNAMIC && !half_width) return DDSP_HIP_EINVAL;
  launch_window_taps(ir, mode, half_width, rows, N, out, S(stream));
  return finish();
}

int ddsp_hip_allpass_backward(const float* c, long ld, long rows, int n_mag, const float* d_re, const float* d_im,
                              float* d_c, void* stream) {
  if (rows < 0 || n_mag < 2 || ld < n_mag) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!c || !d_re || !d_im || !d_c) return DDSP_HIP_EINVAL;
  launch_allpass_backward(c, ld, rows, n_mag, d_re, d_im, d_c, S(stream));
  return finish();
}

int ddsp_hip_fft_convolve(const float* audio, int x_is_u01, const float* taps, const float* addend, float* out,
                          float* out_plain, int B, int F, int hop, int N, int impl, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || N < 2 || (N & 1)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!audio || !taps || !out) return DDSP_HIP_EINVAL;
  int used = launch_fir(audio, x_is_u01, taps, addend, out, out_plain, B, F, hop, N, impl, S(stream));
  if (used < 0) return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_uniform_noise(unsigned long long seed, unsigned long long offset, int B, long T, float* out, void* stream) {
  if (B < 0 || T <= 0) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!out) return DDSP_HIP_EINVAL;
  if (launch_uniform_noise(seed, offset, B, T, out, S(stream)) != 0) return DDSP_HIP_ESHAPE;
  return finish();
}

size_t ddsp_hip_frequency_filter_workspace_bytes(int B, int F, int n_mag) {
  if (B <= 0 || F <= 0 || n_mag < 2) return 0;
  return align_up((size_t)B * F * 2 * (size_t)(n_mag - 1) * sizeof(float), 256);
}

int ddsp_hip_frequency_filter(const float* audio, const float* resp_re, long ld_re, const float* resp_im, long ld_im,
                              int mode, const float* half_width, int B, int F, int hop, int n_mag, const float* table,
                              float* out, void* ws, size_t ws_bytes, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || n_mag < 2 || ld_re < n_mag || (resp_im && ld_im < n_mag)) return DDSP_HIP_EINVAL;
  if (mode < 0 || mode > 2) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!audio || !resp_re || !table || !out || !ws) return DDSP_HIP_EINVAL;
  if (mode == DDSP_HIP_MODE_DYNAMIC && !half_width) return DDSP_HIP_EINVAL;
  if (ws_bytes < ddsp_hip_frequency_filter_workspace_bytes(B, F, n_mag)) return DDSP_HIP_EWS;
  float* taps = static_cast<float*>(ws);
  const long R = (long)B * F;
  const TapsFormScope form;
  synth_taps(resp_re, ld_re, resp_im, ld_im, DDSP_HIP_ACT_NONE, 1.0f, table, mode, half_width, R, n_mag, taps, S(stream));
  if (launch_fir(audio, 0, taps, nullptr, out, nullptr, B, F, hop, 2 * (n_mag - 1), DDSP_HIP_FIR_AUTO, S(stream)) < 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_fft_convolve_backward(const float* audio, int x_is_u01, const float* taps, const float* grad_out,
                                   float* d_audio, float* d_taps, int B, int F, int hop, int N, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || N < 2 || (N & 1)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!audio || !taps || !grad_out || !d_taps) return DDSP_HIP_EINVAL;
  // hop 512, N <= 512: the hop-block form; hop 512, N <= 1022: the per-frame 2048-point form (fir_fft_bwd.hip; knob
  // FIR_BWD_DIRECT = 1: off, for same-box A/Bs); every other shape: direct correlations (fir_bwd_direct.hip)
  if (launch_fir_blk_bwd(audio, x_is_u01, taps, grad_out, d_audio, d_taps, B, F, hop, N, S(stream)) != 0 &&
      (knob(KNOB_FIR_BWD_DIRECT) == 1 ||
       launch_fir_fft_bwd(audio, x_is_u01, taps, grad_out, d_audio, d_taps, B, F, hop, N, S(stream)) != 0) &&
      launch_fir_bwd_direct(audio, x_is_u01, taps, grad_out, d_audio, d_taps, B, F, hop, N, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_combtooth(const float* f0_frames, const float* initial_phase, const double* phase0, int B, int F, int hop,
                       double sr, int infer, float* out, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !phase0 || !out) return DDSP_HIP_EINVAL;
  if (launch_combtooth(f0_frames, initial_phase, B, F, hop, sr, infer, phase0, out, S(stream)) != 0) return DDSP_HIP_EHOP;
  return finish();
}

int ddsp_hip_sinusoid_bank(const float* f0_frames, const float* initial_phase, const double* phase0, const float* c_amp,
                           long ld_amp, int B, int F, int hop, int H, double sr, int infer, float* out, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || H <= 0 || ld_amp < H || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !phase0 || !c_amp || !out) return DDSP_HIP_EINVAL;
  int r = launch_sins_bank(f0_frames, initial_phase, c_amp, ld_amp, B, F, hop, H, sr, infer, phase0, out, S(stream));
  if (r == -1) return DDSP_HIP_EHOP;
  if (r == -2) return DDSP_HIP_ESHAPE;
  return finish();
}

size_t ddsp_hip_sinusoid_bank_backward_scratch_bytes(int B, int F, int H) {
  if (B <= 0 || F <= 0 || H <= 0) return 0;
  return sins_bank_bwd_scratch_floats(B, F, H) * sizeof(float);
}

int ddsp_hip_sinusoid_bank_backward(const float* f0_frames, const float* initial_phase, const double* phase0,
                                    const float* c_amp, long ld_amp, const float* grad_out, int B, int F, int hop, int H,
                                    double sr, int infer, void* scratch, float* d_c_amp, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || H <= 0 || ld_amp < H || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !phase0 || !c_amp || !grad_out || !scratch || !d_c_amp) return DDSP_HIP_EINVAL;
  if (launch_sins_bank_bwd(f0_frames, initial_phase, c_amp, ld_amp, grad_out, B, F, hop, H, sr, infer, phase0,
                           static_cast<float*>(scratch), d_c_amp, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

}  // extern "C"

namespace {

// one synthesiser call as data: what a sub-batch of it is (rows b0 .. b0 + Bn of every per-utterance array)
struct TailCall {
  const float* f0_frames; const float* initial_phase; const double* phase0;
  const float* c0; long ld0;          // Sins: amplitudes            CombSub: group delay
  const float* c1; long ld1;          // Sins: group delay           CombSub: harmonic magnitude
  const float* c2; long ld2;          // noise magnitude
  const float* noise; int noise_is_u01;
  int B, F, hop; double sr; int infer;
  int n0, n1, n2;                     // Sins: H, n_ap, n_nz         CombSub: n_ap, n_harm, n_nz
  const float* t0; const float* t1; const float* t2;       // basis tables (Sins: t0 unused)
  float* signal; float* harmonic; float* noise_out;
  int fir_impl;
  NoiseGen gen;
};

TailCall rows_of(const TailCall& a, int b0, int Bn) {
  TailCall s = a;
  const long r0 = (long)b0 * a.F, t0 = r0 * a.hop;
  s.B = Bn;
  s.f0_frames += r0;
  if (s.initial_phase) s.initial_phase += b0;
  s.phase0 += r0;
  s.c0 += r0 * a.ld0; s.c1 += r0 * a.ld1; s.c2 += r0 * a.ld2;
  if (s.noise) s.noise += t0;
  s.signal += t0;
  if (s.harmonic) s.harmonic += t0;
  if (s.noise_out) s.noise_out += t0;
  s.gen.utt0 = a.gen.utt0 + (unsigned)b0;
  return s;
}

typedef int (*TailFn)(const TailCall&, SynthWs&, hipStream_t, void*);

// Issue `one` over the sub-batches of `a` (lane_plan): sub-batch k on lane k mod 2 with that lane's workspace slot.  Lane 1's
// streams wait for everything the caller's stream holds at entry (the inputs' producers) and are joined back into it before
// the call returns, as the branch stream is: whatever the caller's stream does next is ordered behind the whole call.
int run_lanes(const TailCall& a, const LanePlan& p, void* ws, size_t ws_bytes, int n_max, hipStream_t st, void* aux_stream,
              TailFn one) {
  SynthWs w[2];
  {
    Carver c(ws, ws_bytes);
    for (int l = 0; l < p.slots; ++l) carve_synth(c, p.Bs, a.F, a.hop, n_max, w[l]);
    if (!c.ok) return DDSP_HIP_EWS;
  }
  // a second lane needs a stream pair of its own; without a branch stream (one-stream mode, emulator) the sub-batches run in
  // sequence on the caller's stream, alternating between the slots
  int lane_dev = 0;
  LaneSet* lane1 = (p.slots == 2 && aux_stream && S(aux_stream) != st) ? second_lane(&lane_dev) : nullptr;
  LaneSet* const lane_taken = lane1;                       // goes back to the pool on every path out of this function
  if (lane1) {
    if (hipEventRecord(lane1->fork, st) != hipSuccess || hipStreamWaitEvent(lane1->main1, lane1->fork, 0) != hipSuccess) {
      (void)hipGetLastError();
      lane1 = nullptr;
    }
  }
  int rc = 0;
  t_geometry_batch = a.B;                                // the filters split an utterance into runs as the unsplit call would
  for (int k = 0; k < p.nsub; ++k) {
    const int b0 = k * p.Bs, Bn = a.B - b0 < p.Bs ? a.B - b0 : p.Bs;
    const TailCall sub = rows_of(a, b0, Bn);
    const int l = p.slots == 2 ? (k & 1) : 0;
    const int r = (l == 1 && lane1) ? one(sub, w[1], lane1->main1, lane1->aux1) : one(sub, w[l], st, aux_stream);
    if (r != 0 && rc == 0) rc = r;
    if (rc != 0) break;
  }
  t_geometry_batch = 0;
  if (lane1) {                                           // always joined, also on the error path
    (void)hipEventRecord(lane1->join, lane1->main1);
    (void)hipStreamWaitEvent(st, lane1->join, 0);
  }
  release_lane(lane_taken, lane_dev);
  return rc;
}

// the batch layout of the Sins tail (vocoder.py:580-611) on one (sub-)batch
int sins_rows(const TailCall& a, SynthWs& w, hipStream_t st, void* aux_stream) {
  const int B = a.B, F = a.F, hop = a.hop, H = a.n0, n_ap = a.n1, n_nz = a.n2;
  const long R = (long)B * F;
  float* nz = a.noise_out ? a.noise_out : w.nzbuf;
  // The fused layout (as combsub_rows below; round 6: at every shape): both tap syntheses in one launch -- four dependent
  // launches on the caller's stream: sinusoid bank | taps (grid.y) | noise filter | all-pass filter + noise.  Same kernels, same
  // arguments, same bits as the two-stream layout below.  [MI355X] B = 32 x 10 s, same box: 0.3367 -> 0.3212 ms (r06_v10s_*).
  if (fused_shape_ok(R, F, hop, H, n_ap, n_nz, a.fir_impl, a.gen.on != 0, false)) {
    const int r = launch_sins_bank(a.f0_frames, a.initial_phase, a.c0, a.ld0, B, F, hop, H, a.sr, a.infer, a.phase0, w.buf0, st);
    if (r == -1) return DDSP_HIP_EHOP;
    if (r == -2) return DDSP_HIP_ESHAPE;
    TapsJobs jobs;
    jobs.n = 0;
    const int half = taps_half_ok(F, hop, a.fir_impl) ? 1 : 0;  // the noise filter's taps as half rows (its launch is then the hop-block form's own)
    int ok = launch_taps_pfa510(a.c2, a.ld2, nullptr, 0, 0, DDSP_HIP_ACT_EXP, 1.0f / 128.0f, a.t2, DDSP_HIP_MODE_HANN, nullptr, R,
                                n_nz, w.taps_nz, st, 0.f, &jobs, half);
    ok |= launch_taps_pfa510(a.c1, a.ld1, nullptr, 0, 1, DDSP_HIP_ACT_NONE, 1.0f, a.t1, DDSP_HIP_MODE_ROLL, nullptr, R, n_ap,
                             w.taps, st, 0.f, &jobs);
    if (ok != 0 || launch_taps_pfa510_batch(jobs, st) != 0) return DDSP_HIP_ESHAPE;
    // The two filters as ONE launch: a workgroup runs the noise filter over its run of block pairs and then the all-pass filter over
    // the same run, reading back as addend what it has just stored (k_fir_blk6<.., SEQ>: no second ramp and drain, the same bits).
    // [MI355X] same box: the two launches 68.5 + 70.9 us -> one of ~128 (knob SINS_SEQ = 1: the two launches).  Not with the in-kernel
    // draw, not at other tap counts than the noise filter's (one geometry per launch).
    if (half && !a.gen.on && n_ap == n_nz && knob(KNOB_SINS_SEQ) != 1) {
      const FirSecond second{w.buf0, 0, w.taps, nz, a.signal, a.harmonic, 0, 1};
      if (launch_fir_blk(a.noise, a.noise_is_u01, w.taps_nz, nullptr, nz, nullptr, B, F, hop, 2 * (n_nz - 1), st, nullptr, &second, 1) >= 0)
        return 0;
    }
    if (half) {
      if (launch_fir_blk(a.noise, a.noise_is_u01, w.taps_nz, nullptr, nz, nullptr, B, F, hop, 2 * (n_nz - 1), st, &a.gen, nullptr, 1) < 0)
        return DDSP_HIP_ESHAPE;
    } else if (launch_fir(a.noise, a.noise_is_u01, w.taps_nz, nullptr, nz, nullptr, B, F, hop, 2 * (n_nz - 1), a.fir_impl, st, &a.gen) < 0)
      return DDSP_HIP_ESHAPE;
    if (launch_fir(w.buf0, 0, w.taps, nz, a.signal, a.harmonic, B, F, hop, 2 * (n_ap - 1), a.fir_impl, st) < 0)
      return DDSP_HIP_ESHAPE;
    return 0;
  }
  // noise = Hann-windowed zero-phase filter exp(c)/128 on uniform noise (vocoder.py:603-607), on the second stream
  Branch br(st, aux_stream);           // without a second stream br.aux is the caller's stream: the same launches, in line
  // With the all-pass at 256 bins (prime-factor kernel: no response scratch in the exciter buffer) its taps go to the
  // second stream too, ahead of the noise branch, and the sinusoid bank starts at once (knob STREAM_LAYOUT 1: round-1 order)
  const bool ap_ahead = n_ap == 256 && !t_taps_gemm && knob(KNOB_STREAM_LAYOUT) != 1;
  if (ap_ahead) synth_allpass_taps(a.c1, a.ld1, a.t1, R, n_ap, w.re, w.im, w.taps, br.aux);
  synth_taps(a.c2, a.ld2, nullptr, 0, DDSP_HIP_ACT_EXP, 1.0f / 128.0f, a.t2, DDSP_HIP_MODE_HANN, nullptr, R,
             n_nz, w.taps_nz, br.aux);
  const int rn = launch_fir(a.noise, a.noise_is_u01, w.taps_nz, nullptr, nz, nullptr, B, F, hop, 2 * (n_nz - 1), a.fir_impl,
                            br.aux, &a.gen);
  if (!ap_ahead) synth_allpass_taps(a.c1, a.ld1, a.t1, R, n_ap, w.re, w.im, w.taps, st);
  // exciter: sinusoid bank (vocoder.py:585-594)
  const int r = launch_sins_bank(a.f0_frames, a.initial_phase, a.c0, a.ld0, B, F, hop, H, a.sr, a.infer, a.phase0, w.buf0, st);
  br.join();                                           // always joined, also on the error paths below
  if (r == -1) return DDSP_HIP_EHOP;
  if (r == -2 || rn < 0) return DDSP_HIP_ESHAPE;
  // harmonic = all-pass(sinusoids) (vocoder.py:597-600); signal = harmonic + noise (:609)
  if (launch_fir(w.buf0, 0, w.taps, nz, a.signal, a.harmonic, B, F, hop, 2 * (n_ap - 1), a.fir_impl, st) < 0)
    return DDSP_HIP_ESHAPE;
  return 0;
}

// the batch layout of the CombSub tail (vocoder.py:834-862) on one (sub-)batch
int combsub_rows(const TailCall& a, SynthWs& w, hipStream_t st, void* aux_stream) {
  const int B = a.B, F = a.F, hop = a.hop, n_ap = a.n0, n_harm = a.n1, n_nz = a.n2;
  const long R = (long)B * F;
  const bool all256 = n_ap == 256 && n_harm == 256 && n_nz == 256 && !t_taps_gemm;
  float* nz = a.noise_out ? a.noise_out : w.nzbuf;
  // THE layout of a 256-bin step (round 6): THREE launches on the caller's stream instead of seven on two streams -- exciter and
  // the three tap syntheses (k_front_small, grid.y) | all-pass filter beside the noise filter (grid.y, runs twice as long: half the
  // warm-up passes) | harmonic filter + noise.  Round 4 built it for streaming shapes (B = 1, a fraction of a second per call,
  // gui.py:118-133: the chain of DEPENDENT launches, ~9 us each, is the latency there).  At batch shapes rounds 2 - 5 kept seven
  // launches on two streams "for the overlap" -- but at the clocks' steady state these kernels gain nothing from running beside
  // each other (a filter's three waves per SIMD leave no registers for a fourth wave of anything; one-stream order 0.322 ms,
  // two streams 0.319: what the overlap wins, three cross-stream hand-overs of ~7 us lose), while every separate launch pays its
  // own ramp and drain: the four front kernels 107.6 us one after the other, 81 as one launch; the two independent filters 145 /
  // 125.  [MI355X] same box, B = 32 x 10 s: 0.3194 -> 0.3015 ms (profiles/r06_v8_*, r06_v9_*).  The second stream, its events and
  // the hardware-queue question of round 5 (GPU_MAX_HW_QUEUES) are gone from this path.
  // Same kernels, same arguments as the layouts below (same bits below 4096 frames: tests/test_small_shapes.py; above, the paired
  // filters' run split differs: rounding-level); knob SMALL_PATH = 1: never; knob STREAM_LAYOUT = 1 / 4: not at batch shapes.
  if (w.taps3 && fused_shape_ok(R, F, hop, n_ap, n_harm, n_nz, a.fir_impl, a.gen.on != 0, true)) {
    ExciterJob exc;
    if (make_exciter_job(a.f0_frames, a.initial_phase, B, F, hop, a.sr, a.infer, a.phase0, w.buf0, &exc) != 0) return DDSP_HIP_EHOP;
    TapsJobs jobs;
    jobs.n = 0;
    const int half = taps_half_ok(F, hop, a.fir_impl) ? 1 : 0;  // the noise filter's taps as half rows
    int ok = launch_taps_pfa510(a.c2, a.ld2, nullptr, 0, 0, DDSP_HIP_ACT_EXP, 1.0f / 128.0f, a.t2, DDSP_HIP_MODE_HANN, nullptr, R,
                                n_nz, w.taps_nz, st, 0.f, &jobs, half);
    ok |= launch_taps_pfa510(a.c0, a.ld0, nullptr, 0, 1, DDSP_HIP_ACT_NONE, 1.0f, a.t0, DDSP_HIP_MODE_ROLL, nullptr, R, n_ap,
                             w.taps, st, 0.f, &jobs);
    ok |= launch_taps_pfa510(a.c1, a.ld1, nullptr, 0, 0, DDSP_HIP_ACT_EXP, 1.0f, a.t1, DDSP_HIP_MODE_DYNAMIC, a.f0_frames, R,
                             n_harm, w.taps3, st, (float)a.sr, &jobs);
    if (ok != 0 || launch_taps_pfa510_batch(jobs, st, &exc) != 0) return DDSP_HIP_ESHAPE;   // exciter + the three tap syntheses
    const FirSecond second{a.noise, a.noise_is_u01, w.taps_nz, nullptr, nz, nullptr, half};
    if (launch_fir_blk(w.buf0, 0, w.taps, nullptr, w.buf1, nullptr, B, F, hop, 2 * (n_ap - 1), st, a.gen.on ? &a.gen : nullptr, &second) < 0)
      return DDSP_HIP_ESHAPE;
    if (launch_fir(w.buf1, 0, w.taps3, nz, a.signal, a.harmonic, B, F, hop, 2 * (n_harm - 1), a.fir_impl, st) < 0)
      return DDSP_HIP_ESHAPE;
    return 0;
  }
  Branch br(st, aux_stream);           // without a second stream br.aux is the caller's stream: the same launches, in line
  // Stream layout (knob STREAM_LAYOUT).  1: the noise branch -- its taps and its filter -- on the second stream beside the
  // harmonic chain, joined into the last filter as its addend.  4 (default where every filter has 256 bins; otherwise the
  // all-pass response is staged in the exciter buffer and 1 is used): as 1 with the exciter on the second stream too, ahead
  // of the noise branch, so the (vector-ALU bound) exciter runs beside the (latency bound) all-pass tap synthesis instead
  // of after it.  Same-box A/Bs at B = 32 x 10 s, ms per step: 1 / 4 0.401 / 0.388 (profiles/r02_v14_*); three more that
  // lost (all taps ahead on the second stream, 0.424; the second harmonic filter's taps there too, 0.403; the harmonic
  // chain's front on the second stream, 0.439 against 0.422) are in DESIGN.md section 7 and no longer in the code.
  long layout = knob(KNOB_STREAM_LAYOUT);
  if (layout != 1) layout = 4;
  if (!all256) layout = 1;
  int rc = 0;
  if (layout == 4) {                                       // exciter: combtooth (vocoder.py:839-840)
    rc = launch_combtooth(a.f0_frames, a.initial_phase, B, F, hop, a.sr, a.infer, a.phase0, w.buf0, br.aux);
    br.publish(0);
  }
  // noise branch (vocoder.py:854-858)
  synth_taps(a.c2, a.ld2, nullptr, 0, DDSP_HIP_ACT_EXP, 1.0f / 128.0f, a.t2, DDSP_HIP_MODE_HANN, nullptr, R, n_nz,
             w.taps_nz, br.aux);
  const int rn = launch_fir(a.noise, a.noise_is_u01, w.taps_nz, nullptr, nz, nullptr, B, F, hop, 2 * (n_nz - 1), a.fir_impl,
                            br.aux, &a.gen);
  // all-pass taps (vocoder.py:843-846; at other bin counts their response lives in buf0 until the exciter overwrites it)
  synth_allpass_taps(a.c0, a.ld0, a.t0, R, n_ap, w.re, w.im, w.taps, st);
  if (layout == 4) br.await(0);
  else rc = launch_combtooth(a.f0_frames, a.initial_phase, B, F, hop, a.sr, a.infer, a.phase0, w.buf0, st);
  int r1 = 0;
  if (rc == 0) {
    r1 = launch_fir(w.buf0, 0, w.taps, nullptr, w.buf1, nullptr, B, F, hop, 2 * (n_ap - 1), a.fir_impl, st);
    // harmonic magnitude filter with the f0-dependent window (vocoder.py:847-851); half_width_frames = 1.5 sr / (f0 + 1e-3)
    // (:851) is formed in the kernel's epilogue
    synth_taps(a.c1, a.ld1, nullptr, 0, DDSP_HIP_ACT_EXP, 1.0f, a.t1, DDSP_HIP_MODE_DYNAMIC, a.f0_frames, R, n_harm,
               w.taps, st, (float)a.sr);
  }
  br.join();                                             // always joined, also on the error paths below
  if (rc != 0) return DDSP_HIP_EHOP;
  if (r1 < 0 || rn < 0) return DDSP_HIP_ESHAPE;
  // signal = harmonic + noise (vocoder.py:860): the second harmonic filter adds the branch's result
  if (launch_fir(w.buf1, 0, w.taps, nz, a.signal, a.harmonic, B, F, hop, 2 * (n_harm - 1), a.fir_impl, st) < 0)
    return DDSP_HIP_ESHAPE;
  return 0;
}

}  // namespace

extern "C" {

size_t ddsp_hip_synth_workspace_bytes(int B, int F, int hop, int n_max) {
  if (B <= 0 || F <= 0 || hop <= 0 || n_max < 2) return 0;
  Carver c(nullptr, (size_t)-1);
  c.base = nullptr;
  SynthWs w;
  const LanePlan p = lane_plan(B, F);                    // a split call: one slot per lane, each of one sub-batch
  if (p.nsub == 1) return carve_synth(c, B, F, hop, n_max, w);
  size_t end = 0;
  for (int l = 0; l < p.slots; ++l) end = carve_synth(c, p.Bs, F, hop, n_max, w);
  return end;
}

int ddsp_hip_tail_layout(int combsub, int B, int F, int hop, int n0, int n1, int n2, int fir_impl, int in_kernel_noise,
                         long long offsets[6]) {
  if (!offsets || B <= 0 || F <= 0 || hop <= 0) return DDSP_HIP_EINVAL;
  for (int i = 0; i < 6; ++i) offsets[i] = -1;
  const TapsFormScope form;
  int n_max = n1 > n2 ? n1 : n2;
  if (combsub && n0 > n_max) n_max = n0;
  if (lane_plan(B, F).nsub > 1) return 0;                      // sub-batches on lanes: slots are re-used, nothing survives the call
  if (!fused_shape_ok((long)B * F, F, hop, n0, n1, n2, fir_impl, in_kernel_noise != 0, combsub != 0)) return 0;
  constexpr uintptr_t kBase = 4096;                            // a carve over a pretend base: offsets = pointers - base, null = absent
  Carver c(reinterpret_cast<void*>(kBase), (size_t)1 << 60);
  SynthWs w;
  carve_synth(c, B, F, hop, n_max, w);
  if (combsub && !w.taps3) return 0;
  auto off = [](const float* p) -> long long { return p ? (long long)(reinterpret_cast<uintptr_t>(p) - kBase) : -1; };
  offsets[0] = off(w.buf0);                                    // the exciter [B, T]
  offsets[1] = combsub ? off(w.buf1) : -1;                     // CombSub: the all-pass filter's output [B, T]
  offsets[2] = off(w.taps);                                    // all-pass taps [B, F, N]
  offsets[3] = combsub ? off(w.taps3) : -1;                    // CombSub: harmonic (dynamic-window) taps
  offsets[4] = off(w.taps_nz);                                 // noise taps
  offsets[5] = off(w.nzbuf);                                   // the filtered noise when no noise output was asked for
  return 1;
}

// ---- the backward pass of a fused CombSub tail call (solver.py:93-103: the same forward with gradients) --------------------
// THREE launches on the caller's stream, on the intermediates ddsp_hip_combsub_synth left in ITS workspace (ddsp_hip_tail_layout):
//     k_fir_blk_bwd<true>     harmonic filter: d h1 (input gradient) and d taps_h                  (vocoder.py:847-851 backwards)
//     k_fir_blk_bwd6          all-pass filter's d taps (from d h1)  +  noise filter's d taps        (two jobs)
//     k_taps_pfa510_bwd       the three tap-synthesis adjoints (dynamic window from f0, roll, Hann) (three jobs); the all-pass job
//                             ends in the activation's adjoint, (d re, d im) -> d group-delay control, on its rows in LDS
//                             (knob AP_BWD_SPLIT = 1: k_allpass_backward_256 as a launch of its own, the layout before)
// g_harm / g_noise: the cotangents that reach the harmonic / the noise branch ([B,T]; NULL = none: that branch's gradients are
// not touched).  ws: ddsp_hip_combsub_tail_backward_ws_bytes.
size_t ddsp_hip_combsub_tail_backward_ws_bytes(int B, int F, int hop, int n_mag) {
  if (B <= 0 || F <= 0 || hop <= 0 || n_mag < 2) return 0;
  Carver c(reinterpret_cast<void*>((uintptr_t)4096), (size_t)1 << 60);
  const size_t BT = (size_t)B * F * hop, R = (size_t)B * F, N = 2 * (size_t)(n_mag - 1);
  c.take<float>(BT);
  for (int i = 0; i < 3; ++i) c.take<float>(R * N);
  c.take<float>(R * n_mag);
  c.take<float>(R * n_mag);
  return align_up(c.used, 256);
}

int ddsp_hip_combsub_tail_backward(const float* f0_frames, const float* c_gd, long ld_gd, const float* c_harm, long ld_harm,
                                   const float* c_nz, long ld_nz, const float* noise, int noise_is_u01, const void* fwd_ws,
                                   const float* g_harm, const float* g_noise, int B, int F, int hop, double sr, int n_mag,
                                   const float* table, float* d_gd, float* d_harm, float* d_nz, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (B < 0 || F <= 0 || hop != 512 || n_mag != 256 || !(sr > 0)) return DDSP_HIP_ESHAPE;
  if (B == 0) return 0;
  if (!f0_frames || !c_gd || !c_harm || !c_nz || !noise || !fwd_ws || !table || !ws) return DDSP_HIP_EINVAL;
  if (ld_gd < n_mag || ld_harm < n_mag || ld_nz < n_mag) return DDSP_HIP_EINVAL;
  if ((g_harm && (!d_gd || !d_harm)) || (g_noise && !d_nz)) return DDSP_HIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0 || (reinterpret_cast<uintptr_t>(fwd_ws) & 15) != 0) return DDSP_HIP_EINVAL;
  long long off[6];
  if (ddsp_hip_tail_layout(1, B, F, hop, n_mag, n_mag, n_mag, 0, 0, off) != 1) return DDSP_HIP_ESHAPE;
  const char* fw = static_cast<const char*>(fwd_ws);
  const float* comb = reinterpret_cast<const float*>(fw + off[0]);
  const float* h1 = reinterpret_cast<const float*>(fw + off[1]);
  const float* taps_ap = reinterpret_cast<const float*>(fw + off[2]);
  const float* taps_h = reinterpret_cast<const float*>(fw + off[3]);
  const float* taps_nz = reinterpret_cast<const float*>(fw + off[4]);
  Carver c(ws, ws_bytes);
  const size_t BT = (size_t)B * F * hop, R = (size_t)B * F, N = 2 * (size_t)(n_mag - 1);
  float* d_h1 = c.take<float>(BT);
  float* dt_h = c.take<float>(R * N);
  float* dt_ap = c.take<float>(R * N);
  float* dt_nz = c.take<float>(R * N);
  float* d_re = c.take<float>(R * n_mag);
  float* d_im = c.take<float>(R * n_mag);
  if (!c.ok) return DDSP_HIP_EWS;
  hipStream_t st = S(stream);
  const int Ni = (int)N;
  if (g_harm) {
    if (launch_fir_blk_bwd(h1, 0, taps_h, g_harm, d_h1, dt_h, B, F, hop, Ni, st) != 0) return DDSP_HIP_ESHAPE;
    const FirBwdSecond second{noise, noise_is_u01, g_noise, dt_nz};
    if (launch_fir_blk_bwd(comb, 0, taps_ap, d_h1, nullptr, dt_ap, B, F, hop, Ni, st, g_noise ? &second : nullptr) != 0) {
      // (knob BWD_WPS = 2: the two-wave kernel takes no second job: one launch each)
      if (launch_fir_blk_bwd(comb, 0, taps_ap, d_h1, nullptr, dt_ap, B, F, hop, Ni, st) != 0) return DDSP_HIP_ESHAPE;
      if (g_noise && launch_fir_blk_bwd(noise, noise_is_u01, taps_nz, g_noise, nullptr, dt_nz, B, F, hop, Ni, st) != 0) return DDSP_HIP_ESHAPE;
    }
  } else if (g_noise) {
    if (launch_fir_blk_bwd(noise, noise_is_u01, taps_nz, g_noise, nullptr, dt_nz, B, F, hop, Ni, st) != 0) return DDSP_HIP_ESHAPE;
  }
  TapsBwdJobs jobs;
  jobs.n = 0;
  const bool fuse_ap = (reinterpret_cast<uintptr_t>(d_gd) & 15) == 0 && knob(KNOB_AP_BWD_SPLIT) == 0;
  if (g_harm) {
    jobs.j[jobs.n++] = TapsBwdJob{1, 0, DDSP_HIP_MODE_DYNAMIC, dt_h, c_harm, ld_harm, 1.0f, nullptr, f0_frames, (float)sr, d_harm, nullptr};
    // the all-pass activation's adjoint rides in that job's last stage (no d re / d im round trip, no fourth launch) when the
    // control gradient can take its 16-byte stores
    jobs.j[jobs.n++] = fuse_ap ? TapsBwdJob{0, 1, DDSP_HIP_MODE_ROLL, dt_ap, nullptr, 0, 1.0f, nullptr, nullptr, 0.f, d_re, d_im, c_gd, ld_gd, d_gd}
                               : TapsBwdJob{0, 1, DDSP_HIP_MODE_ROLL, dt_ap, nullptr, 0, 1.0f, nullptr, nullptr, 0.f, d_re, d_im};
  }
  if (g_noise) jobs.j[jobs.n++] = TapsBwdJob{1, 0, DDSP_HIP_MODE_HANN, dt_nz, c_nz, ld_nz, 1.0f / 128.0f, nullptr, nullptr, 0.f, d_nz, nullptr};
  if (jobs.n && launch_taps_pfa510_bwd_jobs(jobs, table, (long)R, st) != 0) return DDSP_HIP_ESHAPE;
  if (g_harm && !fuse_ap) launch_allpass_backward(c_gd, ld_gd, (long)R, n_mag, d_re, d_im, d_gd, st);
  return finish();
}

// d taps [rows, 2 (n_mag - 1)] of the all-pass taps exp(1j cumsum(pi tanh c)) (vocoder.py:581,599 / :834,845 through core.py:254-270)
// -> d c [rows, n_mag]: ddsp_hip_impulse_response_backward (MODE_ROLL, d_re + d_im) followed by ddsp_hip_allpass_backward -- as ONE
// launch at 256 bins (the activation's adjoint in the tap adjoint's last stage: no d re / d im round trip); elsewhere the two
// launches, through d_re_ws / d_im_ws ([rows, n_mag] each; DDSP_HIP_EWS when they are needed and NULL).
int ddsp_hip_allpass_taps_backward(const float* d_taps, const float* c, long ld, long rows, int n_mag, const float* table,
                                   float* d_c, float* d_re_ws, float* d_im_ws, void* stream) {
  if (rows < 0 || n_mag < 2 || ld < n_mag) return DDSP_HIP_EINVAL;
  if (rows == 0) return 0;
  if (!d_taps || !c || !table || !d_c) return DDSP_HIP_EINVAL;
  const TapsFormScope form;
  if (n_mag == 256 && !t_taps_gemm && knob(KNOB_AP_BWD_SPLIT) == 0 && (reinterpret_cast<uintptr_t>(d_c) & 15) == 0) {
    TapsBwdJobs jobs;
    jobs.n = 1;
    jobs.j[0] = TapsBwdJob{0, 1, DDSP_HIP_MODE_ROLL, d_taps, nullptr, 0, 1.0f, nullptr, nullptr, 0.f, nullptr, nullptr, c, ld, d_c};
    if (launch_taps_pfa510_bwd_jobs(jobs, table, rows, S(stream)) == 0) return finish();
  }
  if (!d_re_ws || !d_im_ws) return DDSP_HIP_EWS;
  const int rc = ddsp_hip_impulse_response_backward(d_taps, nullptr, 0, DDSP_HIP_ACT_NONE, 1.0f, DDSP_HIP_MODE_ROLL, nullptr, rows, n_mag,
                                                    table, d_re_ws, d_im_ws, stream);
  if (rc != 0) return rc;
  launch_allpass_backward(c, ld, rows, n_mag, d_re_ws, d_im_ws, d_c, S(stream));
  return finish();
}

int ddsp_hip_sins_synth(const float* f0_frames, const float* initial_phase, const double* phase0, const float* c_amp,
                        long ld_amp, const float* c_gd, long ld_gd, const float* c_nz, long ld_nz, const float* noise,
                        int noise_is_u01, int B, int F, int hop, double sr, int infer, int H, int n_ap, int n_nz,
                        const float* table_ap, const float* table_nz, float* signal, float* harmonic_or_null,
                        float* noise_out_or_null, void* ws, size_t ws_bytes, int fir_impl, void* stream,
                        void* aux_stream, unsigned long long noise_seed, unsigned long long noise_offset) {
  if (B < 0 || F <= 0 || hop <= 0 || H <= 0 || n_ap < 2 || n_nz < 2 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (ld_amp < H || ld_gd < n_ap || ld_nz < n_nz) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !phase0 || !c_amp || !c_gd || !c_nz || !table_ap || !table_nz || !signal)
    return DDSP_HIP_EINVAL;
  // the stream layout below is chosen from "the prime-factor tap synthesis takes these calls", and that kernel stores
  // 16 bytes at a time: the taps carved out of ws must be 16-byte aligned (ddsp_hip.h states the contract)
  if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) return DDSP_HIP_EINVAL;
  // noise == NULL: the uniform draw happens inside the noise filter (philox.h) from (noise_seed, noise_offset)
  const NoiseGen gen{noise_seed, noise_offset, noise ? 0 : 1, 0u};
  if (gen.on && !(hop == 512 && n_nz <= 257 && (fir_impl == 0 || fir_impl == 5))) return DDSP_HIP_ESHAPE;
  const int n_max = n_ap > n_nz ? n_ap : n_nz;
  hipStream_t st = S(stream);
  const TapsFormScope form;
  const TailCall call{f0_frames, initial_phase, phase0, c_amp, ld_amp, c_gd, ld_gd, c_nz, ld_nz, noise, noise_is_u01,
                      B, F, hop, sr, infer, H, n_ap, n_nz, nullptr, table_ap, table_nz, signal, harmonic_or_null,
                      noise_out_or_null, fir_impl, gen};
  const LanePlan plan = lane_plan(B, F);
  if (plan.nsub > 1) {
    const int rc = run_lanes(call, plan, ws, ws_bytes, n_max, st, aux_stream, sins_rows);
    return rc != 0 ? rc : finish();
  }
  Carver c(ws, ws_bytes);
  SynthWs w;
  carve_synth(c, B, F, hop, n_max, w);
  if (!c.ok) return DDSP_HIP_EWS;
  const int rc = sins_rows(call, w, st, aux_stream);
  return rc != 0 ? rc : finish();
}

int ddsp_hip_combsub_synth(const float* f0_frames, const float* initial_phase, const double* phase0, const float* c_gd,
                           long ld_gd, const float* c_harm, long ld_harm, const float* c_nz, long ld_nz,
                           const float* noise, int noise_is_u01, int B, int F, int hop, double sr, int infer, int n_ap,
                           int n_harm, int n_nz, const float* table_ap, const float* table_harm, const float* table_nz,
                           float* signal, float* harmonic_or_null, float* noise_out_or_null, void* ws, size_t ws_bytes,
                           int fir_impl, void* stream, void* aux_stream, unsigned long long noise_seed,
                           unsigned long long noise_offset) {
  if (B < 0 || F <= 0 || hop <= 0 || n_ap < 2 || n_harm < 2 || n_nz < 2 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (ld_gd < n_ap || ld_harm < n_harm || ld_nz < n_nz) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !phase0 || !c_gd || !c_harm || !c_nz || !table_ap || !table_harm || !table_nz || !signal)
    return DDSP_HIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) return DDSP_HIP_EINVAL;      // as ddsp_hip_sins_synth
  const NoiseGen gen{noise_seed, noise_offset, noise ? 0 : 1, 0u};   // noise == NULL: drawn inside the noise filter (philox.h)
  if (gen.on && !(hop == 512 && n_nz <= 257 && (fir_impl == 0 || fir_impl == 5))) return DDSP_HIP_ESHAPE;
  int n_max = n_ap > n_nz ? n_ap : n_nz;
  if (n_harm > n_max) n_max = n_harm;
  hipStream_t st = S(stream);
  const TapsFormScope form;
  const TailCall call{f0_frames, initial_phase, phase0, c_gd, ld_gd, c_harm, ld_harm, c_nz, ld_nz, noise, noise_is_u01,
                      B, F, hop, sr, infer, n_ap, n_harm, n_nz, table_ap, table_harm, table_nz, signal, harmonic_or_null,
                      noise_out_or_null, fir_impl, gen};
  const LanePlan plan = lane_plan(B, F);
  if (plan.nsub > 1) {
    const int rc = run_lanes(call, plan, ws, ws_bytes, n_max, st, aux_stream, combsub_rows);
    return rc != 0 ? rc : finish();
  }
  Carver c(ws, ws_bytes);
  SynthWs w;
  carve_synth(c, B, F, hop, n_max, w);
  if (!c.ok) return DDSP_HIP_EWS;
  const int rc = combsub_rows(call, w, st, aux_stream);
  return rc != 0 ? rc : finish();
}

size_t ddsp_hip_stft_workspace_bytes(int B, int F, int hop) {
  if (B <= 0 || F <= 0 || hop <= 0) return 0;
  return align_up((size_t)B * F * hop * sizeof(float), 256);
}

int ddsp_hip_fast_source(const float* f0_frames, int B, int F, int hop, double sr, float* rad_acc, float* phase_frames,
                         float* combtooth, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !rad_acc) return DDSP_HIP_EINVAL;
  if (launch_fast_source(f0_frames, B, F, hop, sr, rad_acc, phase_frames, combtooth, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_stft_filter(const float* exciter, const float* noise, int noise_is_u01, const float* c_hmag, long ld_hmag,
                         const float* c_hphase, long ld_hphase, const float* c_nmag, long ld_nmag,
                         const float* c_nphase, long ld_nphase, float noise_scale, const float* window, int win,
                         int pad_reflect, int normalize, int B, int F, int hop, float* signal, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || win < 2 || (win & 1)) return DDSP_HIP_EINVAL;
  const int n = win / 2 + 1;
  if (ld_hmag < n || ld_hphase < n || ld_nmag < n || (c_nphase && ld_nphase < n)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!exciter || !noise || !c_hmag || !c_hphase || !c_nmag || !window || !signal) return DDSP_HIP_EINVAL;
  if (pad_reflect && (long)F * hop <= win / 2) return DDSP_HIP_EINVAL;       // reflect padding needs T > win/2
  if (launch_stft_filter(exciter, noise, noise_is_u01, c_hmag, ld_hmag, c_hphase, ld_hphase, c_nmag, ld_nmag, c_nphase,
                         ld_nphase, noise_scale, window, win, pad_reflect, normalize, B, F, hop, signal, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_stft_filter_backward(const float* exciter, const float* noise, int noise_is_u01, const float* c_hmag,
                                  long ld_hmag, const float* c_hphase, long ld_hphase, const float* c_nmag,
                                  long ld_nmag, const float* c_nphase, long ld_nphase, float noise_scale,
                                  const float* window, int win, int pad_reflect, int normalize,
                                  const float* grad_signal, int B, int F, int hop, float* d_hmag, float* d_hphase,
                                  float* d_nmag, float* d_nphase, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || win < 2 || (win & 1)) return DDSP_HIP_EINVAL;
  const int n = win / 2 + 1;
  if (ld_hmag < n || ld_hphase < n || ld_nmag < n || (c_nphase && ld_nphase < n)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!exciter || !noise || !c_hmag || !c_hphase || !c_nmag || !window || !grad_signal || !d_hmag || !d_hphase || !d_nmag)
    return DDSP_HIP_EINVAL;
  if ((c_nphase == nullptr) != (d_nphase == nullptr)) return DDSP_HIP_EINVAL;
  if (pad_reflect && (long)F * hop <= win / 2) return DDSP_HIP_EINVAL;
  if (launch_stft_filter_bwd(exciter, noise, noise_is_u01, c_hmag, ld_hmag, c_hphase, ld_hphase, c_nmag, ld_nmag, c_nphase,
                             ld_nphase, noise_scale, window, win, pad_reflect, normalize, grad_signal, B, F, hop, d_hmag,
                             d_hphase, d_nmag, d_nphase, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_combsubfast_synth(const float* f0_frames, const float* initial_phase, const double* phase0,
                               const float* c_hmag, long ld_hmag, const float* c_hphase, long ld_hphase,
                               const float* c_nmag, long ld_nmag, const float* noise, int noise_is_u01,
                               const float* window, int B, int F, int hop, double sr, int infer, float* signal, void* ws,
                               size_t ws_bytes, void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || !(sr > 0)) return DDSP_HIP_EINVAL;
  const int n = hop + 1;
  if (ld_hmag < n || ld_hphase < n || ld_nmag < n) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !phase0 || !c_hmag || !c_hphase || !c_nmag || !noise || !window || !signal) return DDSP_HIP_EINVAL;
  if (!ws || ws_bytes < ddsp_hip_stft_workspace_bytes(B, F, hop)) return DDSP_HIP_EWS;
  float* comb = static_cast<float*>(ws);
  hipStream_t st = S(stream);
  // exciter: the same combtooth as CombSub (vocoder.py:764-765 == :839-840)
  if (launch_combtooth(f0_frames, initial_phase, B, F, hop, sr, infer, phase0, comb, st) != 0) return DDSP_HIP_EHOP;
  // frames of 2*hop, zero padding, no envelope division (vocoder.py:766-784)
  if (launch_stft_filter(comb, noise, noise_is_u01, c_hmag, ld_hmag, c_hphase, ld_hphase, c_nmag, ld_nmag, nullptr, 0,
                         1.0f / 128.0f, window, 2 * hop, 0, 0, B, F, hop, signal, st) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_combsubsuperfast_synth(const float* f0_frames, const float* rad_acc, const float* c_hmag, long ld_hmag,
                                    const float* c_hphase, long ld_hphase, const float* c_nmag, long ld_nmag,
                                    const float* c_nphase, long ld_nphase, const float* noise, const float* window,
                                    int win, int B, int F, int hop, double sr, float* signal, void* ws, size_t ws_bytes,
                                    void* stream) {
  if (B < 0 || F <= 0 || hop <= 0 || win < 2 || (win & 1) || !(sr > 0)) return DDSP_HIP_EINVAL;
  const int n = win / 2 + 1;
  if (ld_hmag < n || ld_hphase < n || ld_nmag < n || ld_nphase < n) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0_frames || !rad_acc || !c_hmag || !c_hphase || !c_nmag || !c_nphase || !noise || !window || !signal)
    return DDSP_HIP_EINVAL;
  if (!ws || ws_bytes < ddsp_hip_stft_workspace_bytes(B, F, hop)) return DDSP_HIP_EWS;
  float* comb = static_cast<float*>(ws);
  hipStream_t st = S(stream);
  // exciter from the closed-form phase (vocoder.py:643-649); rad_acc was produced before Unit2Control ran
  // (only the per-sample part: the frame-rate scan is the caller's ddsp_hip_fast_source call)
  // torch.stft / istft: reflect padding unless the signal is not longer than win/2 (vocoder.py:667-670)
  const int reflect = (long)F * hop > win / 2 ? 1 : 0;
  // Streaming shapes (gui.py:118-133 runs THIS model, configs/combsub.yaml:19: B = 1, a fraction of a second per call): the
  // exciter is made inside the filter's load path and the tail is ONE launch behind ddsp_hip_fast_source -- a step's latency there
  // is its chain of dependent launches, and a sample recomputed by each of the four frames that cover it costs nothing on an
  // almost empty chip.  Same arithmetic, same bits (tests/test_small_shapes.py); knob SMALL_PATH = 1: the two-launch layout.
  if ((long)B * F < kSmallRows && win == 2048 && hop == 512 && knob(KNOB_SMALL_PATH) != 1) {
    if (launch_stft_filter(nullptr, noise, 0, c_hmag, ld_hmag, c_hphase, ld_hphase, c_nmag, ld_nmag, c_nphase, ld_nphase,
                           1.0f / 128.0f, window, win, reflect, 1, B, F, hop, signal, st, f0_frames, rad_acc, sr) != 0)
      return DDSP_HIP_ESHAPE;
    return finish();
  }
  if (launch_fast_combtooth(f0_frames, rad_acc, B, F, hop, sr, comb, st) != 0) return DDSP_HIP_ESHAPE;
  if (launch_stft_filter(comb, noise, 0, c_hmag, ld_hmag, c_hphase, ld_hphase, c_nmag, ld_nmag, c_nphase, ld_nphase,
                         1.0f / 128.0f, window, win, reflect, 1, B, F, hop, signal, st) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_mel_frames(int T, int n_fft, int hop) {
  if (T < 1 || n_fft < 2 || hop < 1 || hop > n_fft) return DDSP_HIP_EINVAL;
  return mel_frames(T, n_fft, hop);
}

int ddsp_hip_mel_spectrogram(const float* audio, int B, int T, const float* window, int n_fft, int hop,
                             const float* mel_basis, const int* band, const float* band_weights, int n_band_weights,
                             int n_mels, float clip_val, float* out, long stride_b, long stride_mel, long stride_frame,
                             void* stream) {
  if (B < 0 || T < 1 || n_fft < 2 || hop < 1 || hop > n_fft || n_mels < 1) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!audio || !window || !mel_basis || !band || !out) return DDSP_HIP_EINVAL;
  if (launch_mel(audio, B, T, window, n_fft, hop, mel_basis, band, band_weights, n_band_weights, n_mels, clip_val, out,
                 stride_b, stride_mel, stride_frame, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

size_t ddsp_hip_mel_shifted_table_bytes(int n_fft_new, int n_bins) { return mel_czt_table_bytes(n_fft_new, n_bins); }

int ddsp_hip_mel_shifted_tables(int n_fft_new, int win_new, int n_bins, float* tables, void* stream) {
  if (n_fft_new < 2 || win_new < 1 || win_new > n_fft_new || n_bins < 2) return DDSP_HIP_EINVAL;
  if (!mel_czt_table_bytes(n_fft_new, n_bins)) return DDSP_HIP_ESHAPE;
  if (!tables) return DDSP_HIP_EINVAL;
  if (launch_mel_czt_tables(n_fft_new, win_new, n_bins, tables, S(stream)) != 0) return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_mel_shifted_frames(int T, int n_fft_new, int win_new, int hop_new, int center) {
  const int f = mel_czt_frames(T, n_fft_new, win_new, hop_new, center);
  return f < 1 ? DDSP_HIP_EINVAL : f;
}

int ddsp_hip_mel_shifted_spectrogram(const float* audio, int B, int T, const float* tables, int n_fft_new, int win_new,
                                     int hop_new, int center, int n_bins, float mag_scale, const int* band,
                                     const float* band_weights, int n_mels, float clip_val, float* out, long stride_b,
                                     long stride_mel, long stride_frame, void* stream) {
  if (B < 0 || n_mels < 1 || n_bins < 2 || mel_czt_frames(T, n_fft_new, win_new, hop_new, center) < 1) return DDSP_HIP_EINVAL;
  if (!mel_czt_table_bytes(n_fft_new, n_bins)) return DDSP_HIP_ESHAPE;
  if (B == 0) return 0;
  if (!audio || !tables || !band || !band_weights || !out) return DDSP_HIP_EINVAL;
  if (launch_mel_czt(audio, B, T, tables, n_fft_new, win_new, hop_new, center, n_bins, mag_scale, band, band_weights, n_mels,
                     clip_val, out, stride_b, stride_mel, stride_frame, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_sine_source(const float* f0, int B, int L, int upp, double sr, const float* rand_ini, const float* noise,
                         const float* weight, const float* bias, int dim, float sine_amp, float noise_std,
                         float voiced_threshold, float* rad_acc, float* out, void* stream) {
  if (B < 0 || L <= 0 || upp <= 0 || dim <= 0 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0 || !rand_ini || !noise || !weight || !bias || !rad_acc || !out) return DDSP_HIP_EINVAL;
  if (launch_sine_source(f0, B, L, upp, sr, rand_ini, noise, weight, bias, dim, sine_amp, noise_std, voiced_threshold,
                         rad_acc, out, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_sine_source_drawn(const float* f0, int B, int L, int upp, double sr, const float* rand_ini,
                               unsigned long long noise_seed, unsigned long long noise_offset, const float* weight,
                               const float* bias, int dim, float sine_amp, float noise_std, float voiced_threshold,
                               float* rad_acc, float* out, void* stream) {
  if (B < 0 || L <= 0 || upp <= 0 || dim <= 0 || !(sr > 0)) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!f0 || !rand_ini || !weight || !bias || !rad_acc || !out) return DDSP_HIP_EINVAL;
  const NoiseGen gen{noise_seed, noise_offset, 1, 0u};
  if (launch_sine_source(f0, B, L, upp, sr, rand_ini, nullptr, weight, bias, dim, sine_amp, noise_std, voiced_threshold,
                         rad_acc, out, S(stream), &gen) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_normal_noise(unsigned long long seed, unsigned long long offset, int B, long T, int dim, float* out, void* stream) {
  if (B < 0 || T <= 0 || dim <= 0) return DDSP_HIP_EINVAL;
  if (B == 0) return 0;
  if (!out) return DDSP_HIP_EINVAL;
  if (launch_normal_noise(seed, offset, B, T, dim, out, S(stream)) != 0) return DDSP_HIP_ESHAPE;
  return finish();
}

size_t ddsp_hip_spectral_loss_scratch_bytes(int B, long bins_per_utterance) {
  if (B < 1 || bins_per_utterance < 1) return 0;
  return sss_scratch_bytes(B, bins_per_utterance);
}

int ddsp_hip_spectral_loss(const float* spec_true, const float* spec_pred, int B, long bins_per_utterance,
                           float inv_window_norm, float eps, float alpha, void* scratch, size_t scratch_bytes,
                           float* norms, float* loss, void* stream) {
  if (B < 0 || bins_per_utterance < 1 || !(inv_window_norm > 0.f)) return DDSP_HIP_EINVAL;
  if (B == 0) return DDSP_HIP_EINVAL;                                      // a mean over nothing
  if (!spec_true || !spec_pred || !scratch || !norms || !loss) return DDSP_HIP_EINVAL;
  if (scratch_bytes < sss_scratch_bytes(B, bins_per_utterance)) return DDSP_HIP_EWS;
  if (launch_sss_loss(spec_true, spec_pred, B, bins_per_utterance, inv_window_norm, eps, alpha, (double*)scratch, norms,
                      loss, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_spectral_loss_backward(const float* spec_true, const float* spec_pred, int B, long bins_per_utterance,
                                    const float* norms, float inv_window_norm, float eps, float alpha,
                                    const float* grad_out, int wrt_true, float* d_spec, void* stream) {
  if (B < 1 || bins_per_utterance < 1 || !(inv_window_norm > 0.f)) return DDSP_HIP_EINVAL;
  if (!spec_true || !spec_pred || !norms || !grad_out || !d_spec) return DDSP_HIP_EINVAL;
  if (launch_sss_loss_bwd(spec_true, spec_pred, B, bins_per_utterance, norms, inv_window_norm, eps, alpha, grad_out,
                          wrt_true ? 1 : 0, d_spec, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

size_t ddsp_hip_stft_loss_table_bytes(int n_fft) { return czt_table_bytes(n_fft); }

int ddsp_hip_stft_loss_tables(int n_fft, float* tables, void* stream) {
  if (!czt_plan(n_fft)) return DDSP_HIP_ESHAPE;
  if (!tables) return DDSP_HIP_EINVAL;
  if (launch_czt_tables(n_fft, tables, S(stream)) != 0) return DDSP_HIP_ESHAPE;
  return finish();
}

int ddsp_hip_stft_loss_frames(int T, int n_fft, int hop) {
  if (n_fft < 1 || hop < 1 || T < n_fft) return 0;
  return 1 + (T - n_fft) / hop;
}

size_t ddsp_hip_stft_loss_scratch_bytes(int B, int T, int n_fft, int hop) {
  const int frames = ddsp_hip_stft_loss_frames(T, n_fft, hop);
  if (B < 1 || frames < 1 || !czt_plan(n_fft)) return 0;
  return sss_wave_scratch_bytes(B, n_fft, frames);
}

int ddsp_hip_stft_loss(const float* x_true, const float* x_pred, int B, int T, long ld, int n_fft, int hop,
                       const float* tables, float inv_window_norm, float eps, float alpha, void* scratch,
                       size_t scratch_bytes, float* spec_true, float* spec_pred, float* norms, float* loss, void* stream) {
  if (B < 1 || T < 1 || ld < T || !(inv_window_norm > 0.f)) return DDSP_HIP_EINVAL;
  if (!x_true || !x_pred || !tables || !scratch || !spec_true || !spec_pred || !norms || !loss) return DDSP_HIP_EINVAL;
  const int frames = ddsp_hip_stft_loss_frames(T, n_fft, hop);
  if (!czt_plan(n_fft) || frames < 1) return DDSP_HIP_ESHAPE;
  if (scratch_bytes < sss_wave_scratch_bytes(B, n_fft, frames)) return DDSP_HIP_EWS;
  if (launch_sss_wave(x_true, x_pred, B, ld, n_fft, hop, frames, tables, inv_window_norm, eps, alpha, (double*)scratch,
                      spec_true, spec_pred, norms, loss, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

size_t ddsp_hip_stft_loss_backward_ws_bytes(int B, int T, int n_fft, int hop) {
  const int frames = ddsp_hip_stft_loss_frames(T, n_fft, hop);
  if (B < 1 || frames < 1 || !czt_plan(n_fft) || hop > n_fft) return 0;
  return sss_wave_bwd_ws_bytes(B, n_fft, hop, frames);
}

int ddsp_hip_stft_loss_backward(const float* spec_true, const float* spec_pred, int B, int T, int n_fft, int hop,
                                const float* tables, const float* norms, float inv_window_norm, float eps, float alpha,
                                const float* grad_out, int wrt_true, float* d_x, long ld_dx, int accumulate, void* ws,
                                size_t ws_bytes, void* stream) {
  if (B < 1 || T < 1 || ld_dx < T || !(inv_window_norm > 0.f)) return DDSP_HIP_EINVAL;
  if (!spec_true || !spec_pred || !tables || !norms || !grad_out || !d_x) return DDSP_HIP_EINVAL;
  const int frames = ddsp_hip_stft_loss_frames(T, n_fft, hop);
  if (!czt_plan(n_fft) || frames < 1 || hop > n_fft) return DDSP_HIP_ESHAPE;
  const size_t need = sss_wave_bwd_ws_bytes(B, n_fft, hop, frames);
  if (need && (!ws || ws_bytes < need)) return DDSP_HIP_EWS;
  if (need && (reinterpret_cast<uintptr_t>(ws) & 15)) return DDSP_HIP_EINVAL;
  if (launch_sss_wave_bwd(spec_true, spec_pred, B, T, n_fft, hop, frames, tables, norms, inv_window_norm, eps, alpha, grad_out,
                          wrt_true ? 1 : 0, d_x, ld_dx, accumulate ? 1 : 0, (float*)ws, S(stream)) != 0)
    return DDSP_HIP_ESHAPE;
  return finish();
}

}  // extern "C"
