#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
struct Out { float v[12]; };
__host__ __device__ inline void sigma(const float* g, float delta, Out* o) {
    const float w = g[4], x = g[5], y = g[6], z = g[7];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, rx = w * x, ry = w * y, rz = w * z;
    float r[3];
    r[0] = 1.0f - 2.0f * (yy + zz); r[1] = 2.0f * (xy + rz); r[2] = 2.0f * (xz - ry);
    const float kk = delta * g[8];
    const float dx = kk * r[0], dy = kk * r[1], dz = kk * r[2];
    const float wx = g[0] + dx, wy = g[1] + dy, wz = g[2] + dz;
    const float cx = 1.0f * wx + 0.0f * wy + 0.0f * wz + 0.0f;
    const float cz = 0.0f * wx + 0.0f * wy + 1.0f * wz + 4.0f;
    const float un = cx / cz;
    const float ox = un * 128.0f + 64.0f;
    o->v[0] = r[0]; o->v[1] = r[1]; o->v[2] = r[2]; o->v[3] = kk; o->v[4] = dx; o->v[5] = dz; o->v[6] = wx; o->v[7] = wz; o->v[8] = cx; o->v[9] = cz; o->v[10] = un; o->v[11] = ox;
}
__global__ void k(const float* g, float delta, Out* o, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) sigma(g + 12 * i, delta, o + i); }
int main() {
    const int n = 1 << 16;
    std::vector<float> g(12 * n); std::vector<Out> o(n);
    srand(2);
    for (int i = 0; i < n; ++i) { float* p = &g[12 * i]; for (int k = 0; k < 12; ++k) p[k] = (rand() / (float)RAND_MAX) * 2 - 1;
        float nq = sqrtf(p[4]*p[4]+p[5]*p[5]+p[6]*p[6]+p[7]*p[7]); for (int k = 4; k < 8; ++k) p[k] /= nq; p[8] = fabsf(p[8]) * 0.2f + 0.02f; }
    float* dg; Out* dout; (void)hipMalloc(&dg, 48 * n); (void)hipMalloc(&dout, sizeof(Out) * n);
    (void)hipMemcpy(dg, g.data(), 48 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dg, sqrtf(3.0f), dout, n);
    (void)hipMemcpy(o.data(), dout, sizeof(Out) * n, hipMemcpyDeviceToHost);
    int bad[12] = {0};
    for (int i = 0; i < n; ++i) { Out h; sigma(&g[12 * i], sqrtf(3.0f), &h); for (int k = 0; k < 12; ++k) bad[k] += memcmp(&h.v[k], &o[i].v[k], 4) != 0; }
    const char* names[12] = {"r0","r1","r2","kk","dx","dz","wx","wz","cx","cz","un","ox"};
    for (int k = 0; k < 12; ++k) printf("%s:%d ", names[k], bad[k]);
    printf("of %d\n", n);
    return 0;
}
