// sb2.hpp — the two-stage tridiagonalisation of the eigensolver (csrc/sb2.hip), called from syevd.hip.
#pragma once
#include "common.hpp"

namespace pg {

constexpr int SB_B = 64;     // half-width of the band the first stage reduces to = reflector length of the second stage
constexpr int SB_G = 64;     // sweeps per reflector block of the second stage's back-transformation
constexpr int SB_LD = 2 * SB_B;   // row pitch of the compact band storage (band + room for the bulge)
constexpr int SB_MAIL_LD = 512;   // doubles per workgroup mailbox of the stationary bulge-chasing kernel

// work buffers of one solve; every pointer device memory, owned by the caller of sb2_alloc / sb2_free
struct Sb2Work {
    int n = 0, npan = 0, nk = 0, ng = 0, kmax = 0;
    double *Vst = nullptr;    // n x n: stage-1 reflectors where LAPACK keeps them (panel j, column c: rows j + b + c .., unit diagonal explicit)
    double *Tst = nullptr;    // npan x b x b: compact-WY factors of the panels
    double *VW = nullptr;     // n x 2b: [V | W] of the current panel
    double *Qb = nullptr;     // n x b: CholeskyQR intermediate Q1 / A22 V
    double *sm = nullptr;     // 16 x (b x b) small matrices
    double *S = nullptr;      // n x SB_LD compact band
    double *VV = nullptr;     // n x n: stage-2 reflectors, row s = sweep s
    double *TAU = nullptr;    // n x nk
    double *Vp = nullptr, *Vtp = nullptr;   // kmax x ng blocks of 128 x SB_G: parallelogram blocks V and V T
    double *Wws = nullptr;    // (kmax + 1) x SB_G x n
    double *G = nullptr, *T = nullptr, *W = nullptr, *W2 = nullptr;   // stage-1 back-transformation (blocks of 256 reflectors)
    double *mail = nullptr;   // (kmax + 1) x SB_MAIL_LD: mailboxes of the stationary bulge-chasing kernel
    double *Pw = nullptr;     // stage-1 panel work: two transposed copies [V W]' (128 x ldt) + the Gram partials of the fused panel kernels
    int *prog = nullptr;      // n + 16 ints: per-sweep progress | work-queue head | abort flag
    int *fail = nullptr;      // 4 ints: [0] panel factorisation lost orthogonality / not positive definite, [1] bulge-chase wait expired
};

// One solve's work buffers out of ONE device allocation: ~30 hipMalloc / hipFree pairs cost 5 ms per solve (each hipFree waits for the
// device).  While g_arena points at an arena, dev_alloc carves from it (256-byte aligned; falls back to hipMalloc when it is full) and
// dev_free leaves pointers inside it alone; the owner frees the arena's base at the end.
struct DevArena { char *base = nullptr; size_t off = 0, cap = 0; };
extern thread_local DevArena *g_arena;
int dev_alloc(void **p, size_t bytes);
void dev_free(void *p);

size_t sb2_bytes(int n);                 // device memory sb2_alloc takes (for the caller's budget)
int sb2_alloc(int n, Sb2Work &w);
void sb2_free(Sb2Work &w);

// A: n x n fp64 full symmetric (row-major, ld n), destroyed: on return its band |i - j| <= SB_B holds the band matrix
int sy2sb_device(pg_ctx *ctx, int n, double *A, Sb2Work &w);
// band of A -> compact storage -> bulge chasing: d (n), e (n - 1) on the device; reflectors in w.VV / w.TAU
int sb2st_device(pg_ctx *ctx, int n, const double *A, double *d, double *e, Sb2Work &w, bool allow_stationary = true, bool *used_stationary = nullptr);
// Z (n x n row-major) <- Q2 Z, then Z <- Q1 Z
int bt2_prep_device(pg_ctx *ctx, int n, Sb2Work &w, hipStream_t st);   // (V, V T) of every reflector block; st: stream to run on (nullptr: ctx's)
int bt2_device(pg_ctx *ctx, int n, double *Z, Sb2Work &w, bool prepared = false);
int bt1_device(pg_ctx *ctx, int n, double *Z, Sb2Work &w);
void sb2_set_debug(int *host_mapped);   // debugging aid: see pgx_sb2_set_debug

}  // namespace pg
