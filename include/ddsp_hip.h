/* ddsp_hip.h -- C ABI of libddsp_hip.so, the MI355X (gfx950) implementation of the DDSP-SVC
 * harmonic-plus-noise synthesis hot path.
 *
 * The reference (yxlllc/DDSP-SVC) is pure Python: the seam for this path is its Python API
 * (ddsp/core.py functions, ddsp/vocoder.py Sins/CombSub forward).  These entry points are what a
 * ctypes binding on the reference side calls; each one names the reference
 * code it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (torch tensors' data_ptr()), including workspaces -- the library never allocates or frees;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), nothing synchronises;
 *   - return 0 on success, a negative DDSP_HIP_E* code for argument errors, or a positive
 *     hipError_t if a launch failed; ddsp_hip_error_string() decodes both;
 *   - tensors are float32 row-major; B = utterances, F = frames, hop = samples per frame,
 *     T = F*hop; control tensors take a row stride `ld` (floats between consecutive frames) so
 *     the torch.split views of Unit2Control's output can be passed without a copy.
 */
#ifndef DDSP_HIP_H
#define DDSP_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDSP_HIP_VERSION 165          /* 0.1.6.5: + ddsp_hip_allpass_taps_backward; ddsp_hip_combsub_tail_backward is three launches (the all-pass activation's adjoint in the tap adjoint's last stage), knob AP_BWD_SPLIT; 0.1.6.4: the fused layouts keep the noise filter's (even) tap rows as their first half, knob TAPS_FULL; 0.1.6.3: + ddsp_hip_mel_shifted_* (get_mel with keyshift / speed / center, any transform length); 0.1.6.2: + ddsp_hip_tail_layout, ddsp_hip_combsub_tail_backward; the fused one-stream layout of the tails at every shape (knob STREAM_LAYOUT 1 / 4: the two-stream ones); knob BWD_WPS */

#define DDSP_HIP_EINVAL   (-1)        /* bad size / null pointer */
#define DDSP_HIP_EHOP     (-2)        /* hop > 2048: wave-per-frame phase scan does not cover it */
#define DDSP_HIP_ESHAPE   (-3)        /* shape outside what the selected kernel supports */
#define DDSP_HIP_EWS      (-4)        /* workspace too small */

/* window handling of frequency_impulse_response (ddsp/core.py:254-270) */
#define DDSP_HIP_MODE_ROLL    0       /* hann_window=False                     core.py:269 */
#define DDSP_HIP_MODE_HANN    1       /* hann_window=True, half_width=None     core.py:263-264 */
#define DDSP_HIP_MODE_DYNAMIC 2       /* hann_window=True, half_width given    core.py:265-266 */

/* what the response pointers hold */
#define DDSP_HIP_ACT_NONE 0           /* the (real, imaginary) response itself, as ddsp.core takes it */
#define DDSP_HIP_ACT_EXP  1           /* raw control c: response = scale * exp(c)   vocoder.py:580,582,835,836 */

/* FIR implementation selector (0 lets the library choose) */
#define DDSP_HIP_FIR_AUTO   0
#define DDSP_HIP_FIR_SIMPLE 1
#define DDSP_HIP_FIR_MFMA   2       /* 4 waves = 1024 outputs per workgroup */
#define DDSP_HIP_FIR_MFMA8  3       /* 8 waves = 2048 outputs per workgroup */
#define DDSP_HIP_FIR_FFT    4       /* frequency-domain block convolution per frame (2048-point), hop 512, N <= 512 */
#define DDSP_HIP_FIR_BLK    5       /* frequency-domain convolution per hop block (1024-point), hop 512, N <= 512 */

int ddsp_hip_version(void);
const char* ddsp_hip_error_string(int code);

/* Tuning / test knobs of the launchers (workgroup run lengths, which occupancy build of a kernel is launched, ...):
 * names as in DESIGN.md section 7 without the DDSP_HIP_ prefix ("BLK_RUN", "STFT_WPS", ...).  Each knob takes its
 * initial value from the environment variable DDSP_HIP_<NAME> ONCE, at first use; after that only set_tuning changes
 * it (0 = built-in default).  No reference counterpart: measurement tools and the run-split tests use them.
 * Knobs that choose between forms of one operation: TAPS_GEMM (tap synthesis and its adjoint: 0 = by bin count -- prime
 * factors at 256, chirp-z from 112 to 1025, the dense contraction elsewhere; 1 = the dense contraction everywhere;
 * 2 = chirp-z wherever its plans reach), SINS_V1 (sinusoid bank generations), STFT_WPS (waves per SIMD of the short-time
 * spectral filter's variants), CZT_ROUNDS (rounds of resident workgroups of the loss kernels), CZT_TURNS (priority turns of a SIMD's waves in the loss's
 * backward kernel: 0 = on, 2 = off), BLK_WPS (the hop-block filter: 0 / 3 = k_fir_blk6, three waves per SIMD; 2 = round 3's
 * two-wave kernel, kept for same-box A/B runs), SINS_NOSKIP (1 = the sinusoid bank also sums the harmonics that are masked
 * to 1e-7 in both frames of a hop), SMALL_PATH (1 = never take the fused launches of the streaming shapes, B F < 4096: the
 * batch layout at every size; the results are the same bits either way), TAPS_FULL (1 = the fused layouts keep the noise
 * filter's tap rows whole, [B,F,N], instead of their first N/2 + 1 taps -- an even response; same bits either way),
 * AP_BWD_SPLIT (1 = ddsp_hip_combsub_tail_backward runs the all-pass activation's adjoint as a launch of its own instead of
 * in the last stage of the tap adjoint), SINS_SEQ (1 = the Sins tail's two filters as two launches instead of one whose workgroups
 * run the noise filter and then the all-pass filter over the same samples; same bits); the rest are run lengths. */
int ddsp_hip_set_tuning(const char* name, long value);
long ddsp_hip_get_tuning(const char* name);

/* ddsp/core.py:66-70  upsample(signal[B,F,C], factor=hop) -> out[B,F*hop,C] */
int ddsp_hip_upsample(const float* sig, int B, int F, int C, int hop, float* out, void* stream);

/* ddsp/core.py:73-77  remove_above_fmax(amplitudes[rows,H], pitch[rows], fmax, level_start) */
int ddsp_hip_remove_above_fmax(const float* amps, const float* pitch, long rows, int H, float fmax,
                               int level_start, float* out, void* stream);

/* ddsp/vocoder.py:564-575 (Sins) == :819-829 (CombSub): upsample f0, cumulative sum of f0/sr
 * (float64 if infer, float32 outputs of a float64 running sum otherwise), optional initial phase,
 * wrap to [-0.5,0.5].
 *   f0_frames[B,F]; initial_phase[B] radians or NULL;
 *   frame_sums[B,F] (double, scratch); phase0[B,F] (double): unwrapped running sum before each
 *   frame -- the state the synth entry points consume; phase_frames[B,F] = 2*pi*x[:, ::hop]
 *   (what Unit2Control receives); x_or_null[B,T]: the full wrapped phase if the caller wants it. */
int ddsp_hip_phase(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr,
                   int infer, double* frame_sums, double* phase0, float* phase_frames, float* x_or_null,
                   void* stream);

/* basis table for n_mag bins (cosine | sine | periodic Hann); build once per n_mag and keep it */
size_t ddsp_hip_ir_table_bytes(int n_mag);
int ddsp_hip_ir_table(int n_mag, float* table, void* stream);

/* ddsp/vocoder.py:581,599 / :834,845  exp(1j*cumsum(pi*tanh(c), -1)) -> re, im [rows,n_mag] */
int ddsp_hip_allpass_response(const float* c, long ld, long rows, int n_mag, float* re, float* im,
                              void* stream);

/* ddsp/core.py:254-270  frequency_impulse_response: one-sided response [rows,n_mag] (+ imaginary
 * part or NULL) -> causal-form taps [rows, N=2*(n_mag-1)]; `act`/`scale` fuse the control
 * activation, `half_width[rows]` is required for MODE_DYNAMIC. */
int ddsp_hip_impulse_response(const float* resp_re, long ld_re, const float* resp_im, long ld_im, int act,
                              float scale, int mode, const float* half_width, long rows, int n_mag,
                              const float* table, float* taps, void* stream);

/* The two calls above in one, from the raw group-delay control c[rows,n_mag] (row stride ld) to the all-pass taps
 * [rows, 2*(n_mag-1)] (vocoder.py:581,599 / :834,845 + core.py:254-270, hann_window=False).  For n_mag = 256 the response
 * never reaches memory (prime-factor kernel); other sizes stage it in `scratch` (allpass_taps_scratch_bytes). */
size_t ddsp_hip_allpass_taps_scratch_bytes(long rows, int n_mag);
int ddsp_hip_allpass_taps(const float* c, long ld, long rows, int n_mag, const float* table, float* taps, void* scratch,
                          size_t scratch_bytes, void* stream);

/* Adjoints of the two calls above (what autograd returns for the response / the raw control):
 *   impulse_response_backward: d_taps[rows,N] -> d_re[rows,n_mag] (+ d_im[rows,n_mag] when d_im is not NULL:
 *   the complex case, act NONE only; every window mode -- the reference builds its real responses as
 *   torch.complex(param, 0), vocoder.py:606,849,857, so a windowed complex response under autograd is the normal case).  With act EXP the result is the gradient of the raw control
 *   itself (d_re = dL/dc = dL/dresp * scale * exp(c), ctrl/ld_ctrl = the forward's resp_re/ld_re); the window
 *   (mode, half_width) is a constant factor.
 *   allpass_backward: (d_re, d_im)[rows,n_mag] of exp(1j*cumsum(pi*tanh(c))) -> d_c[rows,n_mag]. */
int ddsp_hip_impulse_response_backward(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale,
                                       int mode, const float* half_width, long rows, int n_mag, const float* table,
                                       float* d_re, float* d_im, void* stream);
/* ddsp/core.py:185-237 apply_window_to_impulse_response (window_size = 0, causal = False) and :240-251
 * apply_dynamic_window_to_impulse_response on taps that are already in the time domain: zero-phase taps ir[rows,N] ->
 * windowed causal taps out[rows,N] (out != ir), out[j] = ir[(j - N/2) mod N] * w(j); MODE_ROLL = the bare roll of
 * core.py:269.  N may be odd in MODE_DYNAMIC (the reference allows 2*n_mag-1 there). */
int ddsp_hip_window_impulse_response(const float* ir, int mode, const float* half_width, long rows, int N, float* out,
                                     void* stream);
int ddsp_hip_allpass_backward(const float* c, long ld, long rows, int n_mag, const float* d_re, const float* d_im,
                              float* d_c, void* stream);
/* The two adjoints above back to back for the all-pass taps of vocoder.py:597-600 / :843-846 (what autograd returns for the raw
 * group-delay control c [rows, ld] given d taps [rows, 2 (n_mag - 1)]): d_c [rows, n_mag].  ONE launch at 256 bins (d_c 16-byte
 * aligned); elsewhere two launches through the scratch d_re_ws / d_im_ws ([rows, n_mag] each) -- DDSP_HIP_EWS when they are needed
 * and NULL. */
int ddsp_hip_allpass_taps_backward(const float* d_taps, const float* c, long ld, long rows, int n_mag, const float* table,
                                   float* d_c, float* d_re_ws, float* d_im_ws, void* stream);

/* ddsp/core.py:120-182  fft_convolve(audio[B,T], taps[B,F,N]) -> out[B,T]  (T = F*hop).
 * x_is_u01: the input is a raw U[0,1) draw and 2*u-1 is applied on load (vocoder.py:603,854);
 * addend[B,T] or NULL is added to the result (vocoder.py:609,860); out_plain[B,T] or NULL also
 * receives the un-added result. */
int ddsp_hip_fft_convolve(const float* audio, int x_is_u01, const float* taps, const float* addend,
                          float* out, float* out_plain, int B, int F, int hop, int N, int impl, void* stream);

/* ddsp/core.py:273-280  frequency_filter(audio[B,T], magnitudes[B,F,n_mag] (complex: resp_re + i resp_im, resp_im
 * may be NULL), hann_window, half_width_frames) = fft_convolve(audio, frequency_impulse_response(...)) in one call.
 * mode / half_width[B*F] as ddsp_hip_impulse_response; ws holds the taps
 * (ddsp_hip_frequency_filter_workspace_bytes(B, F, n_mag) bytes); out[B,T]. */
size_t ddsp_hip_frequency_filter_workspace_bytes(int B, int F, int n_mag);
int ddsp_hip_frequency_filter(const float* audio, const float* resp_re, long ld_re, const float* resp_im, long ld_im,
                              int mode, const float* half_width, int B, int F, int hop, int n_mag,
                              const float* table, float* out, void* ws, size_t ws_bytes, void* stream);

/* What autograd returns for ddsp_hip_fft_convolve given grad_out[B,T] = dL/dout: d_taps[B,F,N] and, if
 * d_audio is not NULL, d_audio[B,T] (the adjoints of core.py:120-182; training back-propagates through them,
 * solver.py:93-103).  hop 512, N <= 512: the hop-block FFT form; every other hop / even N: direct correlations
 * (csrc/fir_bwd_direct.hip; ~1 ms per launch at B = 32 x 10 s, N = 1022). */
int ddsp_hip_fft_convolve_backward(const float* audio, int x_is_u01, const float* taps, const float* grad_out,
                                   float* d_audio, float* d_taps, int B, int F, int hop, int N, void* stream);

/* The uniform draw of the noise branch (vocoder.py:603, :854 rand_like) as a counter-based stream, for callers that do
 * not want to materialise torch.rand: u[B,T] in [0,1) from Philox4x32-10 keyed by `seed`, counter = (128 * (t / 512) +
 * t % 128, utterance, offset lo, offset hi), output word (t % 512) / 128 -- reproducible from (seed, offset), independent
 * of launch geometry.  It is NOT torch's stream: the same seed gives different numbers than torch.rand (torch's mapping
 * of elements to counters depends on its launch geometry; reproducing it inside the filter would cost four generator
 * calls per block and thread instead of one).  The synthesiser tails below draw exactly these numbers inside their noise
 * filter when `noise` is NULL (hop 512, n_nz <= 257); this entry point writes them out. */
int ddsp_hip_uniform_noise(unsigned long long seed, unsigned long long offset, int B, long T, float* out, void* stream);

/* DSP tail of Sins.forward (ddsp/vocoder.py:580-611) from raw controls and the phase state.
 * noise[B,T]: uniform draw (noise_is_u01 ? U[0,1) : already 2u-1), or NULL: drawn inside the noise filter from
 * (noise_seed, noise_offset) as ddsp_hip_uniform_noise defines it (opt-in: 4 B / sample less HBM traffic and no separate
 * generator kernel, for ~20 % more arithmetic in that one filter launch); tables for n_ap / n_nz bins.
 * signal[B,T]; harmonic_or_null / noise_out_or_null [B,T] only if the caller wants the tuple. */
int ddsp_hip_sins_synth(const float* f0_frames, const float* initial_phase, const double* phase0,
                        const float* c_amp, long ld_amp, const float* c_gd, long ld_gd,
                        const float* c_nz, long ld_nz, const float* noise, int noise_is_u01,
                        int B, int F, int hop, double sr, int infer, int H, int n_ap, int n_nz,
                        const float* table_ap, const float* table_nz,
                        float* signal, float* harmonic_or_null, float* noise_out_or_null,
                        void* ws, size_t ws_bytes, int fir_impl, void* stream, void* aux_stream,
                        unsigned long long noise_seed, unsigned long long noise_offset);

/* DSP tail of CombSub.forward (ddsp/vocoder.py:834-862). */
int ddsp_hip_combsub_synth(const float* f0_frames, const float* initial_phase, const double* phase0,
                           const float* c_gd, long ld_gd, const float* c_harm, long ld_harm,
                           const float* c_nz, long ld_nz, const float* noise, int noise_is_u01,
                           int B, int F, int hop, double sr, int infer, int n_ap, int n_harm, int n_nz,
                           const float* table_ap, const float* table_harm, const float* table_nz,
                           float* signal, float* harmonic_or_null, float* noise_out_or_null,
                           void* ws, size_t ws_bytes, int fir_impl, void* stream, void* aux_stream,
                           unsigned long long noise_seed, unsigned long long noise_offset);

/* Launch layout (round 6).  With every filter at 256 bins and hop 512 -- the shipped configurations -- a call is issued on `stream`
 * ALONE: CombSub as three launches (exciter + the three tap syntheses as jobs of one launch | all-pass filter + noise filter as two
 * jobs of one launch | harmonic filter + noise), Sins as four; `aux_stream` is not used (independent kernels as jobs of ONE launch save
 * a ramp and a drain each, where two streams paid three cross-stream hand-overs for an overlap that is worth less: 0.319 -> 0.302 ms
 * at B = 32 x 10 s).  Tuning knob STREAM_LAYOUT = 1 / 4 restores the two-stream layouts at batch shapes.  Other bin counts / hops:
 *
 * aux_stream (both calls above): NULL, or a second stream of the same device, which must be the calling thread's
 * current device (hipSetDevice) -- the fork / join events are taken from a per-device pool there.  The noise branch (its taps and its
 * filter: independent of the harmonic chain until the final sum) is then enqueued there -- forked after everything
 * already on `stream`, joined back before the last kernel on `stream`, so for the caller the call still behaves as
 * one operation on `stream` -- and fills the machine where the chain's kernels leave it idle (tails, store phases).
 * Results are bit-identical to the one-stream order.  The events this needs come from a process-wide pool per device, checked
 * out per call (the only objects the library creates, beside the second lane's streams of knob LANE_ROWS). */

/* workspace the two synth entry points need (bytes); n_max = largest n_mag among the filters.  `ws` must be 16-byte
 * aligned (the tap arrays carved out of it are written 16 bytes at a time); an unaligned pointer is DDSP_HIP_EINVAL.
 * 256-bin models at hop 512 (and every model below 4096 frames): the workspace holds one tap buffer more (the fused layout
 * above: all tap syntheses of a step are one launch). */
size_t ddsp_hip_synth_workspace_bytes(int B, int F, int hop, int n_max);

/* Where a ddsp_hip_combsub_synth (combsub != 0; n0, n1, n2 = n_ap, n_harm, n_nz) / ddsp_hip_sins_synth (n0 = H, n1 = n_ap,
 * n2 = n_nz) call of this shape LEAVES ITS INTERMEDIATES in the workspace -- what a training caller keeps for the backward pass
 * (solver.py:93-103: the same forward with gradients) instead of recomputing or re-running the tail as separate operators:
 * byte offsets into `ws` of [0] the exciter [B,T], [1] the all-pass filter's output [B,T] (CombSub), [2] the all-pass taps
 * [B,F,N], [3] the harmonic taps (CombSub), [4] the noise taps -- [B,F,N/2+1]: the first N/2 + 1 taps of an even response (tap
 * N - j is tap j) unless knob TAPS_FULL --, [5] the filtered noise (when no noise output was passed); -1 = not kept.  Returns 1 when the call takes the fused layout and the offsets are valid until the workspace is written again, 0
 * when it does not (other bin counts / hops, in-kernel noise, sub-batch lanes: nothing is promised), < 0 on bad arguments. */
int ddsp_hip_tail_layout(int combsub, int B, int F, int hop, int n0, int n1, int n2, int fir_impl, int in_kernel_noise,
                         long long offsets[6]);

/* The backward pass of a fused ddsp_hip_combsub_synth call (256 / 256 / 256 bins, hop 512: where ddsp_hip_tail_layout returns 1) as
 * THREE launches on `stream`: what autograd returns for the three raw controls of vocoder.py:834-862 given the cotangents that
 * reach the harmonic branch (g_harm [B,T]: d signal + d harmonic) and the noise branch (g_noise [B,T]: d signal + d noise); either
 * may be NULL (that branch's gradients are not written).  fwd_ws is the forward call's workspace, untouched since; f0 / controls /
 * noise are that call's; table = the 256-bin basis table.  d_gd, d_harm, d_nz: [B,F,256] contiguous.  ws (16-byte aligned):
 * ddsp_hip_combsub_tail_backward_ws_bytes(B, F, hop, 256) bytes.  DDSP_HIP_ESHAPE for any other shape (use the per-operator
 * adjoints: ddsp_hip_fft_convolve_backward, ddsp_hip_impulse_response_backward, ddsp_hip_allpass_backward). */
size_t ddsp_hip_combsub_tail_backward_ws_bytes(int B, int F, int hop, int n_mag);
int ddsp_hip_combsub_tail_backward(const float* f0_frames, const float* c_gd, long ld_gd, const float* c_harm, long ld_harm,
                                   const float* c_nz, long ld_nz, const float* noise, int noise_is_u01, const void* fwd_ws,
                                   const float* g_harm, const float* g_noise, int B, int F, int hop, double sr, int n_mag,
                                   const float* table, float* d_gd, float* d_harm, float* d_nz, void* ws, size_t ws_bytes,
                                   void* stream);

/* exciters on their own (used by tests and by callers that want the intermediate):
 * combtooth (vocoder.py:839-840) and the sinusoid bank (vocoder.py:585-594), out[B,T] */
int ddsp_hip_combtooth(const float* f0_frames, const float* initial_phase, const double* phase0, int B, int F,
                       int hop, double sr, int infer, float* out, void* stream);
int ddsp_hip_sinusoid_bank(const float* f0_frames, const float* initial_phase, const double* phase0,
                           const float* c_amp, long ld_amp, int B, int F, int hop, int H, double sr, int infer,
                           float* out, void* stream);

/* Adjoint of ddsp_hip_sinusoid_bank w.r.t. the raw amplitude control: grad_out[B,T] -> d_c_amp[B,F,H]
 * (contiguous).  scratch: ddsp_hip_sinusoid_bank_backward_scratch_bytes(B, F, H) bytes.  Every hop <= 2048 (hop 512: the
 * matrix-pipe form; other hops: one wave per frame, direct sines, as the forward bank there). */
size_t ddsp_hip_sinusoid_bank_backward_scratch_bytes(int B, int F, int H);
int ddsp_hip_sinusoid_bank_backward(const float* f0_frames, const float* initial_phase, const double* phase0,
                                    const float* c_amp, long ld_amp, const float* grad_out, int B, int F, int hop,
                                    int H, double sr, int infer, void* scratch, float* d_c_amp, void* stream);

/* ---- CombSubFast / CombSubSuperFast (ddsp/vocoder.py:613-786): short-time spectral filtering ---- */

/* ddsp/vocoder.py:639-651  CombSubSuperFast.fast_source_gen(f0_frames[B,F]): closed-form per-frame
 * phase with a float32 frame-rate accumulator.
 *   rad_acc[B,F]: fmod(cumsum(rad2), 1) (:646), the state the exciter restarts from;
 *   phase_frames[B,F] or NULL = 2*pi*rad[:, :, 0] (:650, what Unit2Control receives);
 *   combtooth[B,T] or NULL = sinc(rad / (s0 + 1e-5)) (:649). */
int ddsp_hip_fast_source(const float* f0_frames, int B, int F, int hop, double sr, float* rad_acc,
                         float* phase_frames, float* combtooth, void* stream);

/* The shared tail of both models: frames of `win` samples every `hop` (win/2 padding each side:
 * reflect as torch.stft does (:667-684) or zeros (:766)), times window[win], rfft, times the
 * per-frame filters exp(c_hmag + i*pi*c_hphase) (exciter) and noise_scale*exp(c_nmag +
 * i*pi*c_nphase) (noise; c_nphase NULL = zero phase, :760), last filter frame repeated (:662,:759),
 * irfft, times window, overlap-add, crop win/2 (:783-784); normalize != 0 divides by the
 * overlap-added squared window as torch.istft does (:702-708).
 *   exciter, noise, signal [B,T]; controls [B,F,win/2+1] with row strides ld_*; window[win].
 * Supported: hop 512 with win 1024 (CombSubFast) or 2048 (CombSubSuperFast). */
int ddsp_hip_stft_filter(const float* exciter, const float* noise, int noise_is_u01,
                         const float* c_hmag, long ld_hmag, const float* c_hphase, long ld_hphase,
                         const float* c_nmag, long ld_nmag, const float* c_nphase, long ld_nphase,
                         float noise_scale, const float* window, int win, int pad_reflect, int normalize,
                         int B, int F, int hop, float* signal, void* stream);

/* What autograd returns for the four control streams of ddsp_hip_stft_filter given grad_signal[B,T] =
 * dL/dsignal (training: solver.py:93-103 back-propagates through CombSubFast / CombSubSuperFast.forward):
 * d_hmag, d_hphase, d_nmag [B,F,win/2+1] contiguous, d_nphase likewise or NULL when c_nphase is NULL.
 * Same arguments and supported shapes as the forward entry point. */
int ddsp_hip_stft_filter_backward(const float* exciter, const float* noise, int noise_is_u01,
                                  const float* c_hmag, long ld_hmag, const float* c_hphase, long ld_hphase,
                                  const float* c_nmag, long ld_nmag, const float* c_nphase, long ld_nphase,
                                  float noise_scale, const float* window, int win, int pad_reflect, int normalize,
                                  const float* grad_signal, int B, int F, int hop,
                                  float* d_hmag, float* d_hphase, float* d_nmag, float* d_nphase, void* stream);

/* DSP tail of CombSubFast.forward (ddsp/vocoder.py:758-784) from raw controls and the phase state of
 * ddsp_hip_phase: combtooth (:764) -> sqrt-Hann frames of 2*hop -> filters -> overlap-add.
 * noise[B,T]: uniform draw (noise_is_u01 ? U[0,1) : already 2u-1, :771); window[2*hop] is the
 * module's buffer (:726); ws: B*T floats (ddsp_hip_stft_workspace_bytes). */
int ddsp_hip_combsubfast_synth(const float* f0_frames, const float* initial_phase, const double* phase0,
                               const float* c_hmag, long ld_hmag, const float* c_hphase, long ld_hphase,
                               const float* c_nmag, long ld_nmag, const float* noise, int noise_is_u01,
                               const float* window, int B, int F, int hop, double sr, int infer,
                               float* signal, void* ws, size_t ws_bytes, void* stream);

/* DSP tail of CombSubSuperFast.forward (ddsp/vocoder.py:661-708) from raw controls and rad_acc of
 * ddsp_hip_fast_source.  noise[B,T] is the standard-normal draw (:687); window[win] the module's
 * Hann buffer (:629). */
int ddsp_hip_combsubsuperfast_synth(const float* f0_frames, const float* rad_acc,
                                    const float* c_hmag, long ld_hmag, const float* c_hphase, long ld_hphase,
                                    const float* c_nmag, long ld_nmag, const float* c_nphase, long ld_nphase,
                                    const float* noise, const float* window, int win, int B, int F, int hop,
                                    double sr, float* signal, void* ws, size_t ws_bytes, void* stream);

size_t ddsp_hip_stft_workspace_bytes(int B, int F, int hop);

/* ---- log-mel front-end of the cascade (nsf_hifigan/nvSTFT.py:73-117) ---- */

/* STFT.get_mel(y, keyshift=0, speed=1, center=False): pad (win-hop)/2 both sides (reflect, or zeros when
 * the right pad is not shorter than the signal, :97-103), frames of n_fft every hop, window[n_fft] (periodic
 * Hann, :93-94), rfft, sqrt(re^2+im^2+1e-9) (:108), mel_basis[n_mels, n_fft/2+1] @ spec (:115),
 * log(clamp(., clip_val)) (:116).
 *   audio[B,T]; band[n_mels][4] (int32) = {first, one-past-last non-zero bin of the mel_basis row, offset of
 *   the row's band weights in band_weights, 0}: the projection only visits that band; band_weights (or NULL)
 *   holds every row's band back to back (n_band_weights floats, staged on chip when <= 4096, else the dense
 *   basis is read); out element (b, mel, frame) is written at
 *   out[b*stride_b + mel*stride_mel + frame*stride_frame], frames = ddsp_hip_mel_frames(T, n_fft, hop).
 * Supported: n_fft == win == 2048, hop == 512 (the 44.1 kHz NSF-HiFiGAN configuration). */
int ddsp_hip_mel_frames(int T, int n_fft, int hop);
int ddsp_hip_mel_spectrogram(const float* audio, int B, int T, const float* window, int n_fft, int hop,
                             const float* mel_basis, const int* band, const float* band_weights,
                             int n_band_weights, int n_mels, float clip_val,
                             float* out, long stride_b, long stride_mel, long stride_frame, void* stream);

/* STFT.get_mel(y, keyshift, speed, center) in full (nvSTFT.py:73-117; the cascade's formant shift, main_diff.py:359; the
 * pitch augmentation of preprocess.py:88-92), and any (n_fft, win, hop) configuration:
 *   n_fft_new = round(n_fft 2^(keyshift/12)), win_new = round(win 2^(keyshift/12)), hop_new = round(hop speed) (:83-85);
 *   manual padding from (win_new, hop_new) (:97-103); center != 0: torch.stft's reflect padding of n_fft_new/2 on top;
 *   frames of n_fft_new every hop_new, periodic Hann of win_new centred in them; the first min(n_bins, n_fft_new/2 + 1)
 *   bins of the n_fft_new-point DFT -- ANY integer length, a chirp-z transform -- sqrt(re^2+im^2+1e-9) (:108), the bins
 *   above them zero, all times mag_scale (the caller passes win / win_new when keyshift != 0, else 1, :109-114);
 *   mel basis [n_mels, n_bins] as band / band_weights (see above; always read from band_weights); log(clamp(., clip_val)).
 * tables: ddsp_hip_mel_shifted_table_bytes(n_fft_new, n_bins) bytes filled once per (n_fft_new, win_new, n_bins) by
 * ddsp_hip_mel_shifted_tables (the caller caches them per keyshift, as the reference caches its windows, :92-94).
 * Supported: n_bins <= 1025 (n_fft <= 2048); n_fft_new <= 8192 (<= 4096 when min(n_bins, n_fft_new/2 + 1) <= 513: the
 * 2048-point convolution plan); win_new <= n_fft_new, hop_new <= win_new: _table_bytes returns 0 outside.
 * One convolution of 4096 (2048) points per frame up to n_fft_new = 4096 (2048), two beyond.  _frames: the frame count, or
 * DDSP_HIP_EINVAL where torch.stft / F.pad raise (transform longer than the padded signal, center's reflection not
 * shorter than it). */
size_t ddsp_hip_mel_shifted_table_bytes(int n_fft_new, int n_bins);
int ddsp_hip_mel_shifted_tables(int n_fft_new, int win_new, int n_bins, float* tables, void* stream);
int ddsp_hip_mel_shifted_frames(int T, int n_fft_new, int win_new, int hop_new, int center);
int ddsp_hip_mel_shifted_spectrogram(const float* audio, int B, int T, const float* tables, int n_fft_new, int win_new,
                                     int hop_new, int center, int n_bins, float mag_scale, const int* band,
                                     const float* band_weights, int n_mels, float clip_val,
                                     float* out, long stride_b, long stride_mel, long stride_frame, void* stream);

/* ---- harmonic source of NSF-HiFiGAN (nsf_hifigan/models.py:101-204) ---- */

/* SourceModuleHnNSF.forward(f0, upp) = tanh(Linear(SineGen(f0, upp))) with SineGen's two random draws supplied:
 * rand_ini[dim] (initial phase per harmonic, entry 0 must be 0, models.py:150-152) and noise[B, L*upp, dim]
 * (standard normal, models.py:168).  f0[B,L] (0 = unvoiced); weight[dim], bias[1] of the 9 -> 1 linear layer;
 * rad_acc[B,L] scratch; out[B, L*upp].  dim = harmonic_num + 1; supported: 9 (the shipped vocoders) and 1. */
int ddsp_hip_sine_source(const float* f0, int B, int L, int upp, double sr, const float* rand_ini,
                         const float* noise, const float* weight, const float* bias, int dim, float sine_amp,
                         float noise_std, float voiced_threshold, float* rad_acc, float* out, void* stream);

/* The same with the standard-normal noise of models.py:168 (torch.randn_like(sine_waves): [B, L*upp, dim] values, 0.5 GB at
 * B = 32 x 10 s, 1.0 GB at B = 64) DRAWN INSIDE the kernel instead of read: opt-in, a Philox4x32-10 / Box-Muller stream of its
 * own keyed by (noise_seed, noise_offset) -- reproducible, independent of launch geometry, not torch.randn's numbers for a
 * seed.  noise_offset < 2^62, L*upp < 2^32.  ddsp_hip_normal_noise writes the same numbers out as z[B, T, dim]. */
int ddsp_hip_sine_source_drawn(const float* f0, int B, int L, int upp, double sr, const float* rand_ini,
                               unsigned long long noise_seed, unsigned long long noise_offset, const float* weight,
                               const float* bias, int dim, float sine_amp, float noise_std, float voiced_threshold,
                               float* rad_acc, float* out, void* stream);
int ddsp_hip_normal_noise(unsigned long long seed, unsigned long long offset, int B, long T, int dim, float* out,
                          void* stream);

/* ---- spectral loss of the training loop (ddsp/loss.py:9-54) ---- */

/* SSSLoss.forward behind the STFT (loss.py:22-31).  spec_true / spec_pred: the complex STFTs of the two signals
 * (interleaved re, im; any dense layout with the utterance outermost, the SAME layout for both),
 * bins_per_utterance complex values each; S = |X| inv_window_norm + eps (Spectrogram(power=1, normalized=True),
 * loss.py:20); loss[0] = mean_b ||St - Sp||_F / ||St + Sp||_F + alpha mean |log St - log Sp|.  norms[B][2]
 * receives the two Frobenius norms per utterance (input of the backward call); scratch of
 * ddsp_hip_spectral_loss_scratch_bytes(B, bins_per_utterance) bytes.  RSSLoss (loss.py:34-54) calls this once per
 * randomly drawn transform size. */
size_t ddsp_hip_spectral_loss_scratch_bytes(int B, long bins_per_utterance);
int ddsp_hip_spectral_loss(const float* spec_true, const float* spec_pred, int B, long bins_per_utterance,
                           float inv_window_norm, float eps, float alpha, void* scratch, size_t scratch_bytes,
                           float* norms, float* loss, void* stream);
/* d loss / d spec (complex: dRe + i dIm, interleaved, same layout) of the predicted (wrt_true = 0) or the true
 * (1) spectrum, times the upstream gradient grad_out[0] (device scalar). */
int ddsp_hip_spectral_loss_backward(const float* spec_true, const float* spec_pred, int B, long bins_per_utterance,
                                    const float* norms, float inv_window_norm, float eps, float alpha,
                                    const float* grad_out, int wrt_true, float* d_spec, void* stream);

/* The same loss straight from the waveforms, the STFT of Spectrogram(n_fft, hop_length = hop, power = 1,
 * normalized = True, center = False) (loss.py:20) inside the kernel: a chirp-z transform of ANY size
 * 2 <= n_fft <= 2048 (RSSLoss draws them at random, loss.py:47; most have large prime factors), both signals in one
 * complex transform.  tables: ddsp_hip_stft_loss_table_bytes(n_fft) bytes (0: size not supported), filled once per
 * size and device by ddsp_hip_stft_loss_tables.  x_true / x_pred: [B, T] float32 with row stride ld; frames =
 * ddsp_hip_stft_loss_frames(T, n_fft, hop) = 1 + (T - n_fft) / hop (0: signal shorter than one frame).  spec_true /
 * spec_pred receive the complex spectra as [B, frames, n_fft / 2 + 1] (interleaved; the backward call reads them),
 * norms[B][2] and loss[0] as ddsp_hip_spectral_loss.  inv_window_norm = 1 / ||hann(n_fft)||_2. */
size_t ddsp_hip_stft_loss_table_bytes(int n_fft);
int ddsp_hip_stft_loss_tables(int n_fft, float* tables, void* stream);
int ddsp_hip_stft_loss_frames(int T, int n_fft, int hop);
size_t ddsp_hip_stft_loss_scratch_bytes(int B, int T, int n_fft, int hop);
int ddsp_hip_stft_loss(const float* x_true, const float* x_pred, int B, int T, long ld, int n_fft, int hop,
                       const float* tables, float inv_window_norm, float eps, float alpha, void* scratch,
                       size_t scratch_bytes, float* spec_true, float* spec_pred, float* norms, float* loss, void* stream);
/* d loss / d x_pred (wrt_true = 0) or d x_true (1), times grad_out[0]: d_x[B, T] with row stride ld_dx, every sample written
 * (those no frame reaches with 0); accumulate != 0: added to what d_x holds instead (the scales of RSSLoss summed without
 * a temporary each).  hop == n_fft (overlap = 0, the configuration of RSSLoss, loss.py:40): no workspace.  hop < n_fft
 * (overlapping frames): ws of ddsp_hip_stft_loss_backward_ws_bytes(B, T, n_fft, hop) bytes, 16-byte aligned, receives the
 * frames' gradients, which a second kernel gathers per sample in ascending frame order (reproducible). */
size_t ddsp_hip_stft_loss_backward_ws_bytes(int B, int T, int n_fft, int hop);
int ddsp_hip_stft_loss_backward(const float* spec_true, const float* spec_pred, int B, int T, int n_fft, int hop,
                                const float* tables, const float* norms, float inv_window_norm, float eps, float alpha,
                                const float* grad_out, int wrt_true, float* d_x, long ld_dx, int accumulate, void* ws,
                                size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDSP_HIP_H */
