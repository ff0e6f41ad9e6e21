/*
 * CPU oracle (plain C) for the Probing-RAG retrieval-gating hot path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg; never by the product (probing-rag_amd/).
 *
 * Restates, with double accumulators and one final rounding to float:
 *   oracle_prober_forward  <- ImprovedProbe.forward          utils.py:45-57
 *   oracle_gate            <- softmax / sum / threshold      exp_rag.py:407-415
 *   oracle_flat_search     <- faiss.IndexFlatL2/IP.search    utils.py:378-380,
 *                             make_indexer.py:449-457 (faiss-cpu itself is a
 *                             third-party dependency absent from the reference
 *                             tree: definition-level restatement, "parity
 *                             unpinned", ties -> lowest id)
 * Pinned against tests/golden/prober_golden.npz (outputs of the reference's
 * own ImprovedProbe) by tests/test_oracle_c.py.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LN_EPS 1e-5

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void layer_norm(double* v, int n, const float* w, const float* b) {
    double mu = 0.0, var = 0.0;
    for (int i = 0; i < n; ++i) mu += v[i];
    mu /= n;
    for (int i = 0; i < n; ++i) { double t = v[i] - mu; var += t * t; }
    var /= n;                                   /* biased, torch default */
    double rstd = 1.0 / sqrt(var + LN_EPS);
    for (int i = 0; i < n; ++i) v[i] = (v[i] - mu) * rstd * (double)w[i] + (double)b[i];
}

static void linear(const double* in, int nin, const float* W, const float* bias,
                   int nout, double* out) {
    for (int o = 0; o < nout; ++o) {            /* W is [out,in] row-major */
        const float* wr = W + (size_t)o * nin;
        double acc = 0.0;
        for (int i = 0; i < nin; ++i) acc += in[i] * (double)wr[i];
        out[o] = acc + (double)bias[o];
    }
}

static void silu(double* v, int n) {
    for (int i = 0; i < n; ++i) v[i] = v[i] / (1.0 + exp(-v[i]));
}

/* params order: ln0_w ln0_b W1 b1 ln1_w ln1_b W2 b2 ln2_w ln2_b W3 b3 */
int oracle_prober_forward(const float* x, int B, int d, int h, int c,
                          const float* const* params, float* logits) {
#pragma omp parallel
    {
        double* a = (double*)malloc(sizeof(double) * (size_t)(d + 2 * h + c));
        double* h1 = a + d; double* h2 = h1 + h; double* o = h2 + h;
#pragma omp for schedule(static)
        for (int r = 0; r < B; ++r) {
            for (int i = 0; i < d; ++i) a[i] = (double)x[(size_t)r * d + i];
            layer_norm(a, d, params[0], params[1]);          /* utils.py:46 */
            linear(a, d, params[2], params[3], h, h1);       /* :48 */
            silu(h1, h);                                     /* :49 */
            layer_norm(h1, h, params[4], params[5]);         /* :50 (dropout :51 = id) */
            linear(h1, h, params[6], params[7], h, h2);      /* :53 */
            silu(h2, h);                                     /* :54 */
            layer_norm(h2, h, params[8], params[9]);         /* :55 */
            linear(h2, h, params[10], params[11], c, o);     /* :57 */
            for (int j = 0; j < c; ++j) logits[(size_t)r * c + j] = (float)o[j];
        }
        free(a);
    }
    return 0;
}

/* logits [L,B,2] -> probsum [B,2] (float accumulation in layer order, as the
 * reference's CPU tensors do), decision[b] = (s0 + theta < s1) ? 0 : 1 */
int oracle_gate(const float* logits, int L, int B, int ablation, double theta,
                float* probsum, int32_t* decision) {
    for (int b = 0; b < B; ++b) {
        float s0 = 0.f, s1 = 0.f;
        for (int n = ablation; n < L; ++n) {
            double z0 = logits[((size_t)n * B + b) * 2 + 0];
            double z1 = logits[((size_t)n * B + b) * 2 + 1];
            double m = z0 > z1 ? z0 : z1;
            double e0 = exp(z0 - m), e1 = exp(z1 - m);
            s0 += (float)(e0 / (e0 + e1));
            s1 += (float)(e1 / (e0 + e1));
        }
        probsum[2 * b] = s0; probsum[2 * b + 1] = s1;
        decision[b] = ((double)s0 + theta < (double)s1) ? 0 : 1;   /* Python-float compare, exp_rag.py:414 */
    }
    return 0;
}

/* metric: 0 = L2 (squared, ascending), 1 = IP (descending).  COSINE is IP on
 * rows/queries normalised by the caller.  Exact brute force; ties -> lowest id;
 * pads with (+/-FLT_MAX, -1) when N < k. */
int oracle_flat_search(const float* X, int64_t N, int d, const float* Q, int B, int k,
                       int metric, int64_t id_offset, float* D, int64_t* I) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        double* bk = (double*)malloc(sizeof(double) * (size_t)k);
        int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
        int cnt = 0;
        const float* q = Q + (size_t)b * d;
        for (int64_t n = 0; n < N; ++n) {
            const float* xr = X + (size_t)n * d;
            double acc = 0.0;
            if (metric == 0) {
                for (int i = 0; i < d; ++i) { double t = (double)q[i] - (double)xr[i]; acc += t * t; }
            } else {
                for (int i = 0; i < d; ++i) acc += (double)q[i] * (double)xr[i];
                acc = -acc;                       /* smaller key = better */
            }
            if (cnt == k && !(acc < bk[k - 1])) continue;   /* strict: earlier id wins ties */
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && acc < bk[pos - 1]) { bk[pos] = bk[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
            bk[pos] = acc; bi[pos] = n;
            if (cnt < k) ++cnt;
        }
        for (int j = 0; j < k; ++j) {
            if (j < cnt) {
                D[(size_t)b * k + j] = (float)(metric == 0 ? bk[j] : -bk[j]);
                I[(size_t)b * k + j] = bi[j] + id_offset;
            } else {
                D[(size_t)b * k + j] = metric == 0 ? FLT_MAX : -FLT_MAX;
                I[(size_t)b * k + j] = -1;
            }
        }
        free(bk); free(bi);
    }
    return 0;
}
