// All joints of a frame from the full mesh: the body of bf_joints_kernel, also run at the head of the dense keypoint-loss workgroup
// (scan_kernels.hip), which consumes them - one launch less per iteration of the dense schedule.
#pragma once
#include "bf_internal.h"

#define BF_JOINTS_LDS (32 * 3 + 256 * 3 + 4)

// One 256-thread workgroup per frame.  All joints in smplx order: chain joints | selector vertices |
// J_regressor_extra rows (SMPL wrapper, models/smpl.py:72-75) | face landmarks (SMPL-X: 51 static + 17 contour
// landmarks chosen by the neck's yaw), then gathered by joint_map; similarity of smplify.py:189
// applied to the outputs.  `jraw` (optional) receives ALL joints in model space and `lmk_vid` / `lmk_w` the
// vertex ids / barycentric weights of the landmarks actually used, for the dense keypoint loss.
// Body: called by every thread of the workgroup (it synchronises); NT = the workgroup's thread count (a multiple of 256);
// lds = BF_JOINTS_LDS floats of workgroup-shared scratch.
template <int NT>
__device__ __forceinline__ void bf_joints_body(const MeshTab &M, const float *__restrict__ state, const float *__restrict__ vraw,
                                               const float *__restrict__ xpart, float *__restrict__ joints, float *__restrict__ joints_ori,
                                               float *__restrict__ jraw, int *__restrict__ lmk_vid, float *__restrict__ lmk_w,
                                               const int frame, float *lds) {
    float *s_extra = lds, *s_all = lds + 32 * 3;
    int &s_row = *(int *)(lds + 32 * 3 + 256 * 3);
    const int tid = threadIdx.x;
    const int nj = M.nj, nb = M.nb, npf = M.npf, nv = M.nv, ne = M.n_extra, nsel = M.n_selector;
    const int nlm = M.n_lmk_static + M.n_lmk_dyn;
    StateView st = bf_state_view(const_cast<float *>(state) + (size_t)frame * bf_state_stride(nj, npf, nb), nj, npf, nb);
    const float *vr = vraw + (size_t)frame * nv * 3;
    const float t0 = st.t[0], t1 = st.t[1], t2 = st.t[2], sc = st.sc[0] * st.sc[1];
    const int ne3 = ne * 3, nt8 = M.n_tiles;
    // extra-regressor joints: sum the mesh kernel's per-tile partials; 32 lanes per output, each lane's loads issued
    // together (a serial loop over the tiles costs one memory latency per tile), fixed xor tree
    for (int base = 0; base < ne3 * 32; base += NT) {
        const int idx = base + tid, o = idx >> 5, sl = idx & 31;
        float acc = 0.f;
        if (o < ne3) {
            const float *p = xpart + (size_t)frame * nt8 * ne3 + o;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int t = sl + 32 * q; v[q] = t < nt8 ? p[(size_t)t * ne3] : 0.f; }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
            for (int t = sl + 256; t < nt8; t += 32) acc += p[(size_t)t * ne3];
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 8);
        acc += __shfl_xor(acc, 16);
        if (o < ne3 && sl == 0) s_extra[o] = acc;
    }
    if (tid == 0 && M.n_lmk_dyn > 0) {
        // find_dynamic_lmk_idx_and_bcoords: y = round(clamp(-yaw * 180 / pi, max = 39)), negatives folded to 39 - y / 78
        const float *G = st.GR + M.neck_joint * 9;
        float yaw = atan2f(-G[6], sqrtf(G[0] * G[0] + G[3] * G[3]));
        int y = (int)rintf(fminf(-yaw * 180.0f / 3.14159265358979323846f, 39.f));
        if (y < 0) y = y < -39 ? 78 : 39 - y;
        s_row = y;
    }
    __syncthreads();
    const int n_ori = nj + nsel, n_all = n_ori + ne + nlm;
    for (int i = tid; i < n_all * 3; i += NT) {
        int j = i / 3, k = i - j * 3;
        float x;
        if (j < nj) x = st.Gt[j * 3 + k];
        else if (j < n_ori) x = vr[(size_t)M.selector_ids[j - nj] * 3 + k];
        else if (j < n_ori + ne) x = s_extra[(j - n_ori) * 3 + k];
        else {
            int l = j - n_ori - ne;
            const float *bw = l < M.n_lmk_static ? M.lmk_bary + l * 3 : M.dyn_bary + ((size_t)s_row * M.n_lmk_dyn + (l - M.n_lmk_static)) * 3;
            // (the landmark's corner vertices: faces[lmk_faces[l]] / faces[dyn_faces[row][.]], looked up on the host)
            const int *fv = l < M.n_lmk_static ? M.lmk_fv + l * 3 : M.dyn_fv + ((size_t)s_row * M.n_lmk_dyn + (l - M.n_lmk_static)) * 3;
            x = bw[0] * vr[(size_t)fv[0] * 3 + k] + bw[1] * vr[(size_t)fv[1] * 3 + k] + bw[2] * vr[(size_t)fv[2] * 3 + k];
            if (k == 0 && lmk_vid) {
                int *vo = lmk_vid + ((size_t)frame * nlm + l) * 3;
                float *wo = lmk_w + ((size_t)frame * nlm + l) * 3;
                vo[0] = fv[0]; vo[1] = fv[1]; vo[2] = fv[2]; wo[0] = bw[0]; wo[1] = bw[1]; wo[2] = bw[2];
            }
        }
        if (jraw) jraw[(size_t)frame * n_all * 3 + i] = x;
        float tk = k == 0 ? t0 : (k == 1 ? t1 : t2);
        s_all[i] = (x + tk) * sc;
    }
    __syncthreads();
    if (joints_ori)
        for (int i = tid; i < n_ori * 3; i += NT) joints_ori[(size_t)frame * n_ori * 3 + i] = s_all[i];
    if (joints)
        for (int i = tid; i < M.n_joint_map * 3; i += NT)
            joints[(size_t)frame * M.n_joint_map * 3 + i] = s_all[M.joint_map[i / 3] * 3 + i % 3];
}
