#include "../../liverrenderer_amd/csrc/kernels.h"
#include <cstdio>
using namespace lrt;
__global__ void k(DScene sc, const float *in, const int *bi, float *out) {
    int i = threadIdx.x;
    SI si; si.valid = true; si.wi = V3(in[0], in[1], in[2]); si.uv = {in[3], in[4]};
    si.sh.s = V3(1,0,0); si.sh.t = V3(0,1,0); si.sh.n = V3(0,0,1); si.n = V3(0,0,1); si.p = V3(0,0,0); si.dp_du = V3(1,0,0); si.dp_dv = V3(0,1,0);
    BSDFSample bs = bsdf_sample(sc, bi[i], si, in[5], in[6], in[7]);
    float *o = out + 16 * i;
    o[0] = bs.wo.x; o[1] = bs.wo.y; o[2] = bs.wo.z; o[3] = (float) bs.type; o[4] = bs.weight.x; o[5] = bs.eta; o[6] = bs.pdf;
    V3 d = si.sh.to_world(bs.wo); o[7] = d.x; o[8] = d.y; o[9] = d.z;
}
int main() {
    DBsdf hb[3] = { { LRT_BSDF_NULL, -1, -1, -1, 1.f, 1.f, F_NULL, 0 }, { LRT_BSDF_DIELECTRIC, -1, -1, -1, 1.5f, 1.f, F_DELTA, 0 }, { LRT_BSDF_DIFFUSE, 0, -1, -1, 1.f, 1.f, F_SMOOTH, 0 } };
    DTexture ht{}; ht.type = LRT_TEX_RGB; ht.color0[0] = ht.color0[1] = ht.color0[2] = 0.5f;
    DBsdf *db; hipMalloc(&db, sizeof(hb)); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    DTexture *dt; hipMalloc(&dt, sizeof(ht)); hipMemcpy(dt, &ht, sizeof(ht), hipMemcpyHostToDevice);
    float hin[8] = { 0.1f, 0.2f, 0.97f, 0.f, 0.f, 0.5f, 0.3f, 0.7f }; int hbi[3] = { 0, 1, 2 };
    float *din; int *dbi; hipMalloc(&din, 32); hipMalloc(&dbi, 12); hipMemcpy(din, hin, 32, hipMemcpyHostToDevice); hipMemcpy(dbi, hbi, 12, hipMemcpyHostToDevice);
    float *o; hipMalloc(&o, 3 * 64); DScene sc{}; sc.bsdfs = db; sc.textures = dt;
    k<<<1, 3>>>(sc, din, dbi, o);
    float h[48]; hipMemcpy(h, o, 3 * 64, hipMemcpyDeviceToHost);
    for (int t = 0; t < 3; ++t) { for (int i = 0; i < 10; ++i) printf("%g ", h[16 * t + i]); printf("\n"); }
    return 0;
}
