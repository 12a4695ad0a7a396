// host-side launcher prototypes shared between the translation units of libddsp_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "ddsp_common.h"
#include "tuning.h"

namespace ddsp {
// Batch size the filters choose their run length from.  A run's first tap spectrum comes out of a differently packed transform,
// so the LAST BITS of a filter's output depend on how an utterance's block pairs are split into runs, and that split follows B
// (one round of resident workgroups).  A sub-batch of a split call (api.hip, lanes) sets this to the WHOLE call's B, so that
// its samples are those of the unsplit call bit for bit; 0 = the launch's own B.
extern thread_local int t_geometry_batch;
int spl_for_hop(int hop);
void launch_upsample(const float* sig, int B, int F, int C, int hop, float* out, hipStream_t st);
void launch_remove_above_fmax(const float* amps, const float* pitch, long rows, int H, float fmax, int level_start,
                              float* out, hipStream_t st);
int launch_phase(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                 double* frame_sums, double* phase0, float* phase_frames, float* x_or_null, hipStream_t st);
int launch_combtooth(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                     const double* phase0, float* out, hipStream_t st);
int launch_sins_bank(const float* f0_frames, const float* initial_phase, const float* c_amp, long ld_amp, int B, int F,
                     int hop, int H, double sr, int infer, const double* phase0, float* out, hipStream_t st);
size_t sins_bank_bwd_scratch_floats(int B, int F, int H);
int launch_sins_bank_bwd(const float* f0_frames, const float* initial_phase, const float* c_amp, long ld_amp,
                         const float* grad_out, int B, int F, int hop, int H, double sr, int infer, const double* phase0,
                         float* scratch, float* d_c, hipStream_t st);
size_t ir_table_floats(int n);
void launch_ir_table(int n, float* table, hipStream_t st);
void launch_allpass_response(const float* c, long ld, long rows, int n, float* re, float* im, hipStream_t st);
void launch_ir_gemm(const float* a_re, long ld_re, const float* a_im, long ld_im, int act, float scale,
                    const float* table, int mode, const float* half_width, long rows, int n, float* taps,
                    hipStream_t st, float hw_from_f0_sr = 0.f);   // > 0: half_width[] holds f0, width = 1.5 sr / (f0 + 1e-3)
void launch_ir_gemm_bwd(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, const float* table,
                        int mode, const float* half_width, long rows, int n, int has_im, float* d_re, float* d_im,
                        hipStream_t st);
// chirp-z form of the tap synthesis for the other bin counts up to 1025 (ir_czt.hip); 0 = taken, -1 = not its shape.
// hann: the periodic Hann of the basis table (k_ir_table), 2 (n - 1) values
int launch_taps_czt(const float* a_re, long ld_re, const float* a_im, long ld_im, int act, float scale, const float* hann,
                    int mode, const float* half_width, long rows, int n, float* taps, hipStream_t st, float hw_from_f0_sr);
int launch_taps_czt_bwd(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, const float* hann, int mode,
                        const float* half_width, long rows, int n, float* d_re, float* d_im, hipStream_t st);
// prime-factor form of the tap synthesis for n_mag = 256 (ir_pfa.hip); 0 = taken, -1 = not its shape (use launch_ir_gemm).
// allpass_from_control: a_re is the raw group-delay control (pi*tanh -> cumsum -> cos/sin done in the kernel).
// one tap synthesis of k_taps_pfa510 as data: a launch takes up to three of them (grid.y)
struct TapsJob {
  int kind, act, mode;
  const float* a_re; long ld_re;
  const float* a_im; long ld_im;
  float scale;
  const float* hann;
  const float* half_width;
  float hw_sr;
  long rows;
  float* taps;
  int half;                // real kinds under an even window (Hann, none) only: rows of n = N/2 + 1 taps (j <= N/2; tap N - j is tap j)
};
struct TapsJobs { TapsJob j[3]; int n; };
// batch != null: the job is appended to *batch instead of being launched (launch_taps_pfa510_batch launches them together)
int launch_taps_pfa510(const float* a_re, long ld_re, const float* a_im, long ld_im, int allpass_from_control, int act,
                       float scale, const float* table, int mode, const float* half_width, long rows, int n, float* taps,
                       hipStream_t st, float hw_from_f0_sr = 0.f, TapsJobs* batch = nullptr, int half_rows = 0);
// the exciter of a streaming-shape CombSub step as data (launch_combtooth's arguments): rides in the tap launch (k_front_small)
struct ExciterJob {
  const float* f0_frames; const float* initial_phase;
  long n_frames; int F, hop;
  Upsampler up; PhaseCfg cfg;
  const double* phase0; float* out;
};
int make_exciter_job(const float* f0_frames, const float* initial_phase, int B, int F, int hop, double sr, int infer,
                     const double* phase0, float* out, ExciterJob* job);   // 0, or -1 when the shape is not the fused form's
int launch_taps_pfa510_batch(const TapsJobs& jobs, hipStream_t st, const ExciterJob* exciter = nullptr);
int launch_taps_pfa510_bwd(const float* d_taps, const float* ctrl, long ld_ctrl, int act, float scale, const float* table,
                           int mode, const float* half_width, long rows, int n, int has_im, float* d_re, float* d_im,
                           hipStream_t st);
// up to three tap-synthesis adjoints of the same row count as ONE launch (grid.y): a training step's three (n_mag 256).
// hw_sr > 0: half_width holds f0 and the half width is 1.5 hw_sr / (f0 + 1e-3) (vocoder.py:851), formed in the kernel
struct TapsBwdJob {
  int act, has_im, mode;
  const float* d_taps; const float* ctrl; long ld_ctrl; float scale;
  const float* hann; const float* half_width; float hw_sr;
  float* d_re; float* d_im;
  // d_ap != null (a has_im job): the all-pass activation's adjoint (k_allpass_backward_256's arithmetic on the forward kernel's
  // tanh and fixed-point phase) runs in the kernel's last stage on the batch's rows while they are in LDS: the job writes the
  // gradient of the raw group-delay control ap_ctrl [rows, ld_ap] to d_ap [rows, 256] and d_re / d_im are not touched
  const float* ap_ctrl; long ld_ap; float* d_ap;
};
struct TapsBwdJobs { TapsBwdJob j[3]; int n; };
int launch_taps_pfa510_bwd_jobs(const TapsBwdJobs& jobs, const float* table, long rows, hipStream_t st);
void launch_window_taps(const float* in, int mode, const float* half_width, long rows, int N, float* out, hipStream_t st);
void launch_allpass_backward(const float* c, long ld, long rows, int n, const float* d_re, const float* d_im, float* d_c,
                             hipStream_t st);
size_t fir_mfma_lds_bytes(int F, int hop, int N, int waves);
struct NoiseGen;   // philox.h: (seed, offset) of the in-kernel uniform draw
// noise_gen != null && on: the input signal is drawn inside the kernel (hop 512, N <= 512 only: -2 otherwise)
int launch_fir(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
               int B, int F, int hop, int N, int impl, hipStream_t st, const NoiseGen* noise_gen = nullptr);
int launch_uniform_noise(unsigned long long seed, unsigned long long offset, int B, long T, float* out, hipStream_t st);
int launch_fir_fft(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
                   int B, int F, int hop, int N, hipStream_t st);
// second != null: a second, independent filter of the same shape (B, F, hop, N) rides in the same launch (k_fir_blk6, grid.y);
// -1 when this launch cannot take it (the in-kernel noise draw, the two-wave kernel)
// taps_half: the tap rows are the first N/2 + 1 taps of an EVEN response (a zero-phase magnitude filter's under the Hann window:
// tap N - j is tap j), [B, F, N/2 + 1] -- what launch_taps_pfa510(.., half_rows = 1) writes; the fused layouts of api.hip keep the
// noise filter's taps so.  (NOT the harmonic filter's: the dynamic window clamps its upper side only, core.py:245, and is not even.)
// seq: the second filter runs BEHIND the first in the same workgroups (one row of them, the one-job launch's run split) and may take
// the first one's result as its addend (k_fir_blk6<.., SEQ>); -1 with the in-kernel draw
struct FirSecond { const float* x; int x_is_u01; const float* taps; const float* addend; float* out; float* out_plain; int taps_half; int seq = 0; };
int launch_fir_blk(const float* x, int x_is_u01, const float* taps, const float* addend, float* out, float* out_plain,
                   int B, int F, int hop, int N, hipStream_t st, const NoiseGen* noise_gen = nullptr,
                   const FirSecond* second = nullptr, int taps_half = 0);
// second != null: a second, independent TAP gradient of the same shape rides in the same launch (k_fir_blk_bwd6, grid.y); -1 when
// this launch cannot take it (an input gradient is wanted, knob BWD_WPS = 2)
struct FirBwdSecond { const float* x; int x_is_u01; const float* grad_out; float* d_taps; };
int launch_fir_blk_bwd(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                       int B, int F, int hop, int N, hipStream_t st, const FirBwdSecond* second = nullptr);
// hop 512, even N <= 1022 (fir_fft_bwd.hip): the per-frame 2048-point form, what N = 514 .. 1022 take; d_x may be null
int launch_fir_fft_bwd(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                       int B, int F, int hop, int N, hipStream_t st);
// every hop / tap count (fir_bwd_direct.hip): direct correlations, behind launch_fir_blk_bwd as k_fir_simple is behind the forward forms
int launch_fir_bwd_direct(const float* x, int x_is_u01, const float* taps, const float* grad_out, float* d_x, float* d_taps,
                          int B, int F, int hop, int N, hipStream_t st);
int launch_fast_source(const float* f0_frames, int B, int F, int hop, double sr, float* rad_acc, float* phase_frames,
                       float* combtooth, hipStream_t st);
int launch_fast_combtooth(const float* f0_frames, const float* rad_acc, int B, int F, int hop, double sr, float* out,
                          hipStream_t st);
int launch_stft_filter(const float* exc, const float* noise, int noise_is_u01, const float* c_hmag, long ld_hm,
                       const float* c_hphase, long ld_hp, const float* c_nmag, long ld_nm, const float* c_nphase,
                       long ld_np, float noise_scale, const float* window, int win, int reflect, int normalize, int B,
                       int F, int hop, float* out, hipStream_t st,
                       // streaming shapes of CombSubSuperFast: the exciter made in the filter's load path from (f0_frames, rad_acc, sr)
                       const float* exc_f0 = nullptr, const float* exc_acc = nullptr, double exc_sr = 0.0);
int launch_stft_filter_bwd(const float* exc, const float* noise, int noise_is_u01, const float* c_hmag, long ld_hm,
                           const float* c_hphase, long ld_hp, const float* c_nmag, long ld_nm, const float* c_nphase,
                           long ld_np, float noise_scale, const float* window, int win, int reflect, int normalize,
                           const float* grad_out, int B, int F, int hop, float* d_hmag, float* d_hphase, float* d_nmag,
                           float* d_nphase, hipStream_t st);
int launch_mel(const float* audio, int B, int T, const float* window, int n_fft, int hop, const float* basis,
               const int* band, const float* packed, int packed_len, int n_mels, float clip, float* out, long sb,
               long sm, long sf, hipStream_t st);
int mel_frames(int T, int n_fft, int hop);
// the same front-end at any transform length / hop / centring (mel_czt.hip; nvSTFT.py:83-85,109-114)
int mel_czt_plan(int n_new, int n_bins);
size_t mel_czt_table_bytes(int n_new, int n_bins);
int mel_czt_frames(int T, int n_new, int win_new, int hop_new, int center);
int launch_mel_czt_tables(int n_new, int win_new, int n_bins, float* tab, hipStream_t st);
int launch_mel_czt(const float* audio, int B, int T, const float* tab, int n_new, int win_new, int hop_new, int center,
                   int n_bins, float mag_scale, const int* band, const float* packed, int n_mels, float clip, float* out,
                   long sb, long sm, long sf, hipStream_t st);
// gen != null && on: the standard-normal noise is drawn inside the kernel from (seed, offset) (noise may be null)
int launch_sine_source(const float* f0, int B, int L, int upp, double sr, const float* rand_ini, const float* noise,
                       const float* weight, const float* bias, int dim, float sine_amp, float noise_std,
                       float voiced_threshold, float* rad_acc, float* out, hipStream_t st, const NoiseGen* gen = nullptr);
int launch_normal_noise(unsigned long long seed, unsigned long long offset, int B, long T, int dim, float* out, hipStream_t st);
int sss_chunks(int B, long per_utt);
size_t sss_scratch_bytes(int B, long per_utt);
int launch_sss_loss(const float* xt, const float* xp, int B, long per_utt, float inv_wn, float eps, float alpha,
                    double* scratch, float* norms, float* loss, hipStream_t st);
int launch_sss_loss_bwd(const float* xt, const float* xp, int B, long per_utt, const float* norms, float inv_wn,
                        float eps, float alpha, const float* grad_out, int wrt_true, float* dx, hipStream_t st);
void launch_sss_final(const double* scratch, int B, int chunks, long per_utt, float alpha, float* norms, float* loss,
                      hipStream_t st);
int czt_plan(int n);
size_t czt_table_bytes(int n);
int launch_czt_tables(int n, float* tab, hipStream_t st);
size_t sss_wave_scratch_bytes(int B, int n, int frames);
int launch_sss_wave(const float* xt, const float* xp, int B, long ld, int n, int hop, int frames, const float* tab,
                    float inv_wn, float eps, float alpha, double* scratch, float* spec_t, float* spec_p, float* norms,
                    float* loss, hipStream_t st);
size_t sss_wave_bwd_ws_bytes(int B, int n, int hop, int frames);
int launch_sss_wave_bwd(const float* spec_t, const float* spec_p, int B, int T, int n, int hop, int frames, const float* tab,
                        const float* norms, float inv_wn, float eps, float alpha, const float* grad_out, int wrt_true,
                        float* dx, long ld_dx, int accumulate, float* ws, hipStream_t st);
}  // namespace ddsp
