// numerics probe: are f32 '/', sqrtf, and a*b+c (contract off) bit-identical to IEEE host results on gfx950?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const float* a, const float* b, const float* c, float* q, float* s, float* m, float* w, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = a[i] / b[i];
    s[i] = sqrtf(fabsf(a[i]));
    m[i] = a[i] * b[i] + c[i];
    float u = a[i] / (b[i] + 4.0f);
    w[i] = u * 128.0f + 64.0f;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), c(n), q(n), s(n), m(n), w(n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = (rand() / (float)RAND_MAX) * 2 - 1; b[i] = (rand() / (float)RAND_MAX) * 2 - 1; c[i] = (rand() / (float)RAND_MAX) * 2 - 1; }
    float *da, *db, *dc, *dq, *ds, *dm, *dw;
    hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dc, 4 * n); hipMalloc(&dq, 4 * n); hipMalloc(&ds, 4 * n); hipMalloc(&dm, 4 * n); hipMalloc(&dw, 4*n);
    hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 4 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dc, dq, ds, dm, dw, n);
    hipMemcpy(q.data(), dq, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(w.data(), dw, 4 * n, hipMemcpyDeviceToHost);
    int bq = 0, bs = 0, bm = 0, bw = 0;
    for (int i = 0; i < n; ++i) {
        volatile float hq = a[i] / b[i]; volatile float hs = sqrtf(fabsf(a[i])); volatile float p = a[i] * b[i]; volatile float hm = p + c[i];
        volatile float u = a[i] / (b[i] + 4.0f); volatile float u2 = u * 128.0f; volatile float hw = u2 + 64.0f;
        float x;
        x = hq; bq += memcmp(&x, &q[i], 4) != 0; x = hs; bs += memcmp(&x, &s[i], 4) != 0; x = hm; bm += memcmp(&x, &m[i], 4) != 0; x = hw; bw += memcmp(&x, &w[i], 4) != 0;
    }
    printf("mismatch div %d sqrt %d muladd %d chain %d of %d\n", bq, bs, bm, bw, n);
    return 0;
}
