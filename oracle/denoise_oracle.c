/*
 * TEST INFRASTRUCTURE — CPU oracle #2 for the denoise forward, plain C, no dependencies.
 *
 * A restatement, from the definitions of the operators, of the reference's
 * DenoiseGenerator (reference backend/app.py:39-103): fp32 NCHW tensors, the reference's own
 * weight layouts (Conv2d [Cout,Cin,3,3], ConvTranspose2d [Cin,Cout,2,2]) and parameter order.
 * Every accumulation is a plain (unfused) multiply-add in a fixed (ci, kh, kw) order, in float
 * (acc64 = 0: the reference's arithmetic type) or in double with fp32 storage between layers
 * (acc64 = 1: used to rank fp32 implementations by their distance from the exact value).
 *
 * Parity pinned: tests/test_oracle.py checks this against tests/golden/ fixtures generated from
 * the reference class itself (tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped package never links or calls it.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ftree-vectorize -mavx2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_BAD_SHAPE 1
#define ORACLE_NOMEM 2

/* y = act(conv3x3(x, w, b)), stride 1, zero pad 1.   reference app.py:43-77 (nn.Conv2d 3x3 p1) */
static int conv3x3(const float* in, const float* w, const float* b, float* out,
                   int N, int Ci, int Co, int H, int W, int relu, int acc64)
{
    int fail = 0;
#pragma omp parallel
    {
        const size_t plane = (size_t)H * W;
        float* accf = (float*)malloc(plane * sizeof(float));
        double* accd = acc64 ? (double*)malloc(plane * sizeof(double)) : NULL;
        if (!accf || (acc64 && !accd)) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for collapse(2) schedule(dynamic, 1)
            for (int n = 0; n < N; ++n)
                for (int co = 0; co < Co; ++co) {
                    if (acc64) for (size_t i = 0; i < plane; ++i) accd[i] = (double)b[co];
                    else       for (size_t i = 0; i < plane; ++i) accf[i] = b[co];
                    for (int ci = 0; ci < Ci; ++ci) {
                        const float* ip = in + ((size_t)n * Ci + ci) * plane;
                        const float* wp = w + ((size_t)co * Ci + ci) * 9;
                        for (int kh = 0; kh < 3; ++kh)
                            for (int kw = 0; kw < 3; ++kw) {
                                const float wv = wp[kh * 3 + kw];
                                const int y0 = kh == 0 ? 1 : 0, y1 = kh == 2 ? H - 1 : H;
                                const int x0 = kw == 0 ? 1 : 0, x1 = kw == 2 ? W - 1 : W;
                                for (int y = y0; y < y1; ++y) {
                                    const float* ir = ip + (size_t)(y + kh - 1) * W + (kw - 1);
                                    if (acc64) {
                                        double* ar = accd + (size_t)y * W;
                                        const double wd = (double)wv;
                                        for (int x = x0; x < x1; ++x) ar[x] += wd * (double)ir[x];
                                    } else {
                                        float* ar = accf + (size_t)y * W;
                                        for (int x = x0; x < x1; ++x) ar[x] += wv * ir[x];
                                    }
                                }
                            }
                    }
                    float* op = out + ((size_t)n * Co + co) * plane;
                    for (size_t i = 0; i < plane; ++i) {
                        float v = acc64 ? (float)accd[i] : accf[i];
                        op[i] = (relu && !(v > 0.0f)) ? 0.0f : v;   /* nn.ReLU: max(x, 0) */
                    }
                }
        }
        free(accf);
        free(accd);
    }
    return fail ? ORACLE_NOMEM : ORACLE_OK;
}

/* nn.MaxPool2d(2, 2), floor mode.   reference app.py:48,56 */
static void maxpool2(const float* in, float* out, int NC, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < NC; ++p) {
        const float* ip = in + (size_t)p * H * W;
        float* op = out + (size_t)p * Ho * Wo;
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                const float* q = ip + (size_t)(2 * y) * W + 2 * x;
                float m = q[0];
                if (q[1] > m) m = q[1];
                if (q[W] > m) m = q[W];
                if (q[W + 1] > m) m = q[W + 1];
                op[(size_t)y * Wo + x] = m;
            }
    }
}

/* nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2): weight [Ci,Co,2,2], no activation.
 * out[n,co,2y+kh,2x+kw] = b[co] + sum_ci in[n,ci,y,x] * w[ci,co,kh,kw].   reference app.py:65,73 */
static int convT2x2(const float* in, const float* w, const float* b, float* out,
                    int N, int Ci, int Co, int H, int W, int acc64)
{
    int fail = 0;
#pragma omp parallel
    {
        const size_t plane = (size_t)H * W;
        float* accf = (float*)malloc(plane * sizeof(float));
        double* accd = acc64 ? (double*)malloc(plane * sizeof(double)) : NULL;
        if (!accf || (acc64 && !accd)) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for collapse(2) schedule(dynamic, 1)
            for (int n = 0; n < N; ++n)
                for (int co = 0; co < Co; ++co)
                    for (int k = 0; k < 4; ++k) {
                        if (acc64) for (size_t i = 0; i < plane; ++i) accd[i] = (double)b[co];
                        else       for (size_t i = 0; i < plane; ++i) accf[i] = b[co];
                        for (int ci = 0; ci < Ci; ++ci) {
                            const float* ip = in + ((size_t)n * Ci + ci) * plane;
                            const float wv = w[((size_t)ci * Co + co) * 4 + k];
                            if (acc64) { const double wd = wv; for (size_t i = 0; i < plane; ++i) accd[i] += wd * (double)ip[i]; }
                            else       for (size_t i = 0; i < plane; ++i) accf[i] += wv * ip[i];
                        }
                        const int kh = k >> 1, kw = k & 1;
                        float* op = out + ((size_t)n * Co + co) * plane * 4;
                        for (int y = 0; y < H; ++y)
                            for (int x = 0; x < W; ++x)
                                op[(size_t)(2 * y + kh) * (2 * W) + 2 * x + kw] =
                                    acc64 ? (float)accd[(size_t)y * W + x] : accf[(size_t)y * W + x];
                    }
        }
        free(accf);
        free(accd);
    }
    return fail ? ORACLE_NOMEM : ORACLE_OK;
}

/* torch.cat([up, skip[:, :, :Hu, :Wu]], dim=1): upsampled first, skip cropped top-left.
 * reference app.py:90-93, 97-100 */
static void cat_crop(const float* up, int Cu, const float* skip, int Cs, int Hs, int Ws,
                     float* out, int N, int Hu, int Wu)
{
    const size_t plane = (size_t)Hu * Wu;
    for (int n = 0; n < N; ++n) {
        float* o = out + (size_t)n * (Cu + Cs) * plane;
        memcpy(o, up + (size_t)n * Cu * plane, (size_t)Cu * plane * sizeof(float));
        for (int c = 0; c < Cs; ++c)
            for (int y = 0; y < Hu; ++y)
                memcpy(o + ((size_t)(Cu + c)) * plane + (size_t)y * Wu,
                       skip + (((size_t)n * Cs + c) * Hs + y) * Ws, (size_t)Wu * sizeof(float));
    }
}

/* Parameter offsets into the flat blob: reference state_dict order (app.py:42-78),
 * weight then bias per layer. */
static const int kCi[12] = {3, 64, 64, 128, 128, 256, 256, 256, 128, 128, 128, 64};
static const int kCo[12] = {64, 64, 128, 128, 256, 256, 128, 128, 128, 64, 64, 3};
static const int kT[12]  = {0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0};   /* 1 = ConvTranspose2d */

size_t cid_oracle_param_count(void)
{
    size_t n = 0;
    for (int l = 0; l < 12; ++l) n += (size_t)kCi[l] * kCo[l] * (kT[l] ? 4 : 9) + kCo[l];
    return n;   /* 1,827,587 */
}

/* Output spatial size for an HxW input: 4*floor(H/4) x 4*floor(W/4)  (two floor pools, two x2 ups). */
int cid_oracle_out_hw(int H, int W, int* Ho, int* Wo)
{
    if (H < 4 || W < 4) return ORACLE_BAD_SHAPE;
    *Ho = 4 * (H / 4);
    *Wo = 4 * (W / 4);
    return ORACLE_OK;
}

/*
 * Whole forward.  params: flat fp32 blob in state_dict order.  x: [N,3,H,W].  out: [N,3,Ho,Wo].
 * stages: NULL, or 9 caller-allocated buffers that receive the outputs of
 * down1, pool1, down2, pool2, bottleneck, up2, upconv2, up1, upconv1 (pre-tanh), NCHW.
 * reference DenoiseGenerator.forward, app.py:80-103.
 */
int cid_oracle_forward(const float* params, const float* x, float* out,
                       int N, int H, int W, int acc64, float** stages)
{
    if (N < 1 || H < 4 || W < 4) return ORACLE_BAD_SHAPE;
    const float* wp[12];
    const float* bp[12];
    {
        const float* p = params;
        for (int l = 0; l < 12; ++l) {
            wp[l] = p; p += (size_t)kCi[l] * kCo[l] * (kT[l] ? 4 : 9);
            bp[l] = p; p += kCo[l];
        }
    }
    const int H1 = H / 2, W1 = W / 2, H2 = H1 / 2, W2 = W1 / 2;
    const int Hu2 = 2 * H2, Wu2 = 2 * W2, Hu1 = 2 * Hu2, Wu1 = 2 * Wu2;
    const size_t s0 = (size_t)N * H * W, s1 = (size_t)N * H1 * W1, s2 = (size_t)N * H2 * W2;
    const size_t su2 = (size_t)N * Hu2 * Wu2, su1 = (size_t)N * Hu1 * Wu1;
    int rc = ORACLE_OK;
    float *t0 = malloc(s0 * 64 * 4), *e1 = malloc(s0 * 64 * 4), *p1 = malloc(s1 * 64 * 4);
    float *t1 = malloc(s1 * 128 * 4), *e2 = malloc(s1 * 128 * 4), *p2 = malloc(s2 * 128 * 4);
    float *t2 = malloc(s2 * 256 * 4), *bt = malloc(s2 * 256 * 4);
    float *u2 = malloc(su2 * 128 * 4), *c2 = malloc(su2 * 256 * 4), *t3 = malloc(su2 * 128 * 4), *d2 = malloc(su2 * 128 * 4);
    float *u1 = malloc(su1 * 64 * 4), *c1 = malloc(su1 * 128 * 4), *t4 = malloc(su1 * 64 * 4), *d1 = malloc(su1 * 3 * 4);
    if (!t0 || !e1 || !p1 || !t1 || !e2 || !p2 || !t2 || !bt || !u2 || !c2 || !t3 || !d2 || !u1 || !c1 || !t4 || !d1) {
        rc = ORACLE_NOMEM;
        goto done;
    }
#define TRY(e) do { rc = (e); if (rc) goto done; } while (0)
    TRY(conv3x3(x, wp[0], bp[0], t0, N, 3, 64, H, W, 1, acc64));              /* down1      app.py:81 */
    TRY(conv3x3(t0, wp[1], bp[1], e1, N, 64, 64, H, W, 1, acc64));
    maxpool2(e1, p1, N * 64, H, W);                                           /* pool1      app.py:82 */
    TRY(conv3x3(p1, wp[2], bp[2], t1, N, 64, 128, H1, W1, 1, acc64));         /* down2      app.py:84 */
    TRY(conv3x3(t1, wp[3], bp[3], e2, N, 128, 128, H1, W1, 1, acc64));
    maxpool2(e2, p2, N * 128, H1, W1);                                        /* pool2      app.py:85 */
    TRY(conv3x3(p2, wp[4], bp[4], t2, N, 128, 256, H2, W2, 1, acc64));        /* bottleneck app.py:87 */
    TRY(conv3x3(t2, wp[5], bp[5], bt, N, 256, 256, H2, W2, 1, acc64));
    TRY(convT2x2(bt, wp[6], bp[6], u2, N, 256, 128, H2, W2, acc64));          /* up2        app.py:89 */
    cat_crop(u2, 128, e2, 128, H1, W1, c2, N, Hu2, Wu2);                      /*            app.py:90-93 */
    TRY(conv3x3(c2, wp[7], bp[7], t3, N, 256, 128, Hu2, Wu2, 1, acc64));      /* upconv2    app.py:94 */
    TRY(conv3x3(t3, wp[8], bp[8], d2, N, 128, 128, Hu2, Wu2, 1, acc64));
    TRY(convT2x2(d2, wp[9], bp[9], u1, N, 128, 64, Hu2, Wu2, acc64));         /* up1        app.py:96 */
    cat_crop(u1, 64, e1, 64, H, W, c1, N, Hu1, Wu1);                          /*            app.py:97-100 */
    TRY(conv3x3(c1, wp[10], bp[10], t4, N, 128, 64, Hu1, Wu1, 1, acc64));     /* upconv1    app.py:101 */
    TRY(conv3x3(t4, wp[11], bp[11], d1, N, 64, 3, Hu1, Wu1, 0, acc64));
    for (size_t i = 0; i < su1 * 3; ++i)                                      /* tanh       app.py:103 */
        out[i] = acc64 ? (float)tanh((double)d1[i]) : tanhf(d1[i]);
    if (stages) {
        memcpy(stages[0], e1, s0 * 64 * 4);   memcpy(stages[1], p1, s1 * 64 * 4);
        memcpy(stages[2], e2, s1 * 128 * 4);  memcpy(stages[3], p2, s2 * 128 * 4);
        memcpy(stages[4], bt, s2 * 256 * 4);  memcpy(stages[5], u2, su2 * 128 * 4);
        memcpy(stages[6], d2, su2 * 128 * 4); memcpy(stages[7], u1, su1 * 64 * 4);
        memcpy(stages[8], d1, su1 * 3 * 4);
    }
done:
    free(t0); free(e1); free(p1); free(t1); free(e2); free(p2); free(t2); free(bt);
    free(u2); free(c2); free(t3); free(d2); free(u1); free(c1); free(t4); free(d1);
    return rc;
}
