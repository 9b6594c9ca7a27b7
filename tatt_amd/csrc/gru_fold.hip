// GruBlock = 1x1 conv followed by nn.GRU (model/tsrn.py:1082-1092) with NO non-linearity in between, so the GRU's input
// projection of both directions composes with the conv into one matrix per block and step:
//     gi = W_ih (W_c x + b_c) + b_ih = (W_ih W_c) x + (W_ih b_c + b_ih)            W_ih = [forward; reverse] stacked, [192][64]
// tatt_gru_fold_w builds Wf = W_ih W_c [192][Cin] and bf [192] for up to FOLD_MAX blocks in one launch (per training step, beside
// the conv weight pack); tatt_gru_unfold_grad maps the gradient of (Wf, bf) back onto the reference's parameters:
//     dW_ih = dWf W_c^T + dbf b_c^T     db_ih = dbf     dW_c = W_ih^T dWf     db_c = W_ih^T dbf
// The [R,64] intermediate of the 1x1 conv (16 MB per block at B = 64) and its GEMMs never exist.
#include "common.h"

namespace {

constexpr int FOLD_MAX = 16;
struct FoldGroup {
  const float* Wc[FOLD_MAX];      // [64][Cin]
  const float* bc[FOLD_MAX];      // [64]
  const float* Wih[FOLD_MAX][2];  // [96][64] forward, reverse
  const float* bih[FOLD_MAX][2];  // [96]
  float* Wf[FOLD_MAX];            // [192][Cin], followed by WcT [Cin][64] (the conv weight transposed, for the gradient kernel)
  float* bf[FOLD_MAX];            // [192]
  int Cin[FOLD_MAX];
};

__global__ __launch_bounds__(256) void gru_fold_w_k(FoldGroup g) {
  const int b = blockIdx.y, Cin = g.Cin[b];
  const int i = blockIdx.x * 256 + threadIdx.x;  // (j, c) of Wf, then j of bf
  if (i < 192 * Cin) {
    const int j = i / Cin, c = i - j * Cin;
    const float* wi = g.Wih[b][j >= 96] + (j % 96) * 64;
    const float* wc = g.Wc[b] + c;
    float s = 0.f;
#pragma unroll 8
    for (int m = 0; m < 64; ++m) s += wi[m] * wc[(long)m * Cin];
    g.Wf[b][i] = s;
  } else if (i < 192 * Cin + 192) {
    const int j = i - 192 * Cin;
    const float* wi = g.Wih[b][j >= 96] + (j % 96) * 64;
    float s = g.bih[b][j >= 96][j % 96];
    for (int m = 0; m < 64; ++m) s += wi[m] * g.bc[b][m];
    g.bf[b][j] = s;
  } else if (i < 192 * Cin + 192 + 64 * Cin) {  // WcT[c][m] = Wc[m][c]
    const int e = i - 192 * Cin - 192, c = e >> 6, m = e & 63;
    g.Wf[b][192 * Cin + e] = g.Wc[b][(long)m * Cin + c];
  }
}

struct UnfoldArgs {
  const float* dWf;  // [192][Cin]
  const float* dbf;  // [192]
  const float* WcT;  // [Cin][64] (written by tatt_gru_fold_w behind Wf)
  const float* bc;   // [64]
  const float* Wih[2];
  float* dWih[2];  // [96][64] +=
  float* dbih[2];  // [96] +=
  float* dWc;      // [64][Cin] +=
  float* dbc;      // [64] +=
  int Cin;
};

struct UnfoldGroup {
  UnfoldArgs u[FOLD_MAX];
};

__device__ __forceinline__ void gru_unfold_body(const UnfoldArgs& u);

__global__ __launch_bounds__(256) void gru_unfold_grad_k(UnfoldArgs u) { gru_unfold_body(u); }
// blockIdx.y = block of the group: the ten GruBlocks of a backward pass in ONE launch (each is a latency-bound handful of workgroups)
__global__ __launch_bounds__(256) void gru_unfold_grad_group_k(UnfoldGroup g) { gru_unfold_body(g.u[blockIdx.y]); }

__device__ __forceinline__ void gru_unfold_body(const UnfoldArgs& u) {
  const int Cin = u.Cin;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 192 * 64) {  // dW_ih[j][m] += sum_c dWf[j][c] Wc[m][c] + dbf[j] bc[m]
    const int j = i >> 6, m = i & 63;
    const float* df = u.dWf + (long)j * Cin;   // wave-uniform row: scalar loads
    const float* wc = u.WcT + m;                // consecutive lanes read consecutive floats
    float s0 = u.dbf[j] * u.bc[m], s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
    for (int c = 0; c < Cin; c += 4) {          // four independent chains, eight trips unrolled: 64 loads in flight per thread
      s0 += df[c] * wc[(long)c * 64];
      s1 += df[c + 1] * wc[(long)(c + 1) * 64];
      s2 += df[c + 2] * wc[(long)(c + 2) * 64];
      s3 += df[c + 3] * wc[(long)(c + 3) * 64];
    }
    u.dWih[j >= 96][(j % 96) * 64 + m] += (s0 + s1) + (s2 + s3);
    return;
  }
  i -= 192 * 64;
  if (i < 192) {
    u.dbih[i >= 96][i % 96] += u.dbf[i];
    return;
  }
  i -= 192;
  if (i < 64 * Cin) {  // dW_c[m][c] += sum_j W_ih[j][m] dWf[j][c]
    const int m = i / Cin, c = i - m * Cin;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float* wi = u.Wih[d] + m;
      const float* df = u.dWf + (long)d * 96 * Cin + c;
#pragma unroll 8
      for (int j = 0; j < 96; j += 4) {
        s0 += wi[j * 64] * df[(long)j * Cin];
        s1 += wi[(j + 1) * 64] * df[(long)(j + 1) * Cin];
        s2 += wi[(j + 2) * 64] * df[(long)(j + 2) * Cin];
        s3 += wi[(j + 3) * 64] * df[(long)(j + 3) * Cin];
      }
    }
    u.dWc[i] += (s0 + s1) + (s2 + s3);
    return;
  }
  i -= 64 * Cin;
  if (i < 64) {
    float s = 0.f;
    for (int j = 0; j < 192; ++j) s += u.Wih[j >= 96][(j % 96) * 64 + i] * u.dbf[j];
    u.dbc[i] += s;
  }
}

}  // namespace

extern "C" {

// n <= 16 GruBlocks in one launch; all array arguments are HOST arrays of length n (Wih / bih: 2n entries, forward then reverse per block).
int tatt_gru_fold_w(int n, const float* const* Wc, const float* const* bc, const float* const* Wih, const float* const* bih,
                    float* const* Wf, float* const* bf, const int* Cin, hipStream_t st) {
  TATT_CHECK_ARG(n > 0 && n <= FOLD_MAX && Wc && bc && Wih && bih && Wf && bf && Cin, "tatt_gru_fold_w: 1..%d blocks, non-null arrays", FOLD_MAX);
  FoldGroup g;
  memset(&g, 0, sizeof(g));
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    TATT_CHECK_ARG(Wc[i] && bc[i] && Wih[2 * i] && Wih[2 * i + 1] && bih[2 * i] && bih[2 * i + 1] && Wf[i] && bf[i] && Cin[i] > 0,
                   "tatt_gru_fold_w: bad block %d", i);
    g.Wc[i] = Wc[i]; g.bc[i] = bc[i]; g.Wf[i] = Wf[i]; g.bf[i] = bf[i]; g.Cin[i] = Cin[i];
    for (int d = 0; d < 2; ++d) { g.Wih[i][d] = Wih[2 * i + d]; g.bih[i][d] = bih[2 * i + d]; }
    if (Cin[i] > mx) mx = Cin[i];
  }
  hipLaunchKernelGGL(gru_fold_w_k, dim3(cdiv(192 * mx + 192 + 64 * mx, 256), n), dim3(256), 0, st, g);
  TATT_LAUNCH_CHECK("gru_fold_w");
  return TATT_OK;
}

int tatt_gru_unfold_grad(const float* dWf, const float* dbf, const float* WcT, const float* bc, const float* Wih_f, const float* Wih_b,
                         float* dWih_f, float* dWih_b, float* dbih_f, float* dbih_b, float* dWc, float* dbc, int Cin, hipStream_t st) {
  TATT_CHECK_ARG(dWf && dbf && WcT && bc && Cin % 4 == 0 && Wih_f && Wih_b && dWih_f && dWih_b && dbih_f && dbih_b && dWc && dbc && Cin > 0,
                 "tatt_gru_unfold_grad: bad arguments");
  UnfoldArgs u{dWf, dbf, WcT, bc, {Wih_f, Wih_b}, {dWih_f, dWih_b}, {dbih_f, dbih_b}, dWc, dbc, Cin};
  hipLaunchKernelGGL(gru_unfold_grad_k, dim3(cdiv(192 * 64 + 192 + 64 * Cin + 64, 256)), dim3(256), 0, st, u);
  TATT_LAUNCH_CHECK("gru_unfold_grad");
  return TATT_OK;
}

// The same for n <= 16 GruBlocks in one launch.  HOST arrays of length n (Wih / dWih / dbih: 2n entries, forward then reverse per block).
int tatt_gru_unfold_grad_group(int n, const float* const* dWf, const float* const* dbf, const float* const* WcT, const float* const* bc,
                               const float* const* Wih, float* const* dWih, float* const* dbih, float* const* dWc, float* const* dbc,
                               const int* Cin, hipStream_t st) {
  TATT_CHECK_ARG(n > 0 && n <= FOLD_MAX && dWf && dbf && WcT && bc && Wih && dWih && dbih && dWc && dbc && Cin,
                 "tatt_gru_unfold_grad_group: 1..%d blocks, non-null arrays", FOLD_MAX);
  UnfoldGroup g;
  memset(&g, 0, sizeof(g));
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    TATT_CHECK_ARG(dWf[i] && dbf[i] && WcT[i] && bc[i] && Wih[2 * i] && Wih[2 * i + 1] && dWih[2 * i] && dWih[2 * i + 1] && dbih[2 * i] &&
                       dbih[2 * i + 1] && dWc[i] && dbc[i] && Cin[i] > 0 && Cin[i] % 4 == 0, "tatt_gru_unfold_grad_group: bad block %d", i);
    g.u[i] = UnfoldArgs{dWf[i], dbf[i], WcT[i], bc[i], {Wih[2 * i], Wih[2 * i + 1]}, {dWih[2 * i], dWih[2 * i + 1]},
                        {dbih[2 * i], dbih[2 * i + 1]}, dWc[i], dbc[i], Cin[i]};
    if (Cin[i] > mx) mx = Cin[i];
  }
  hipLaunchKernelGGL(gru_unfold_grad_group_k, dim3(cdiv(192 * 64 + 192 + 64 * mx + 64, 256), n), dim3(256), 0, st, g);
  TATT_LAUNCH_CHECK("gru_unfold_grad_group");
  return TATT_OK;
}

}  // extern "C"
