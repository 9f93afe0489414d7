// C ABI: projection matcher (include/sind_hip.h, "sind_match_*").
#include <cstring>
#include "../../include/sind_hip.h"
#include "match.hpp"

struct sind_match {
    int device = 0, maxB = 0; sind::MatchParams prm{}; float mb = 0; hipStream_t stream = nullptr;
    DevBuf<sind::MatchPose> pose; DevBuf<int> nLast, nCur, lastOct, curOct, gstart, gidx, choice, minOwner, matchOfCur, nmatches, rounds;
    DevBuf<float> x3Dw, lastAng, curXY, curAng, curUR; DevBuf<uint8_t> lastFlags, curTaken; DevBuf<uint32_t> lastDesc, curDesc;
    // host staging (one H2D per array and call)
    std::vector<sind::MatchPose> h_pose; std::vector<int> h_nLast, h_nCur, h_lastOct, h_curOct, h_gstart, h_gidx, h_match, h_nm, h_rounds;
    std::vector<float> h_x3Dw, h_lastAng, h_curXY, h_curAng, h_curUR; std::vector<uint8_t> h_lastFlags, h_curTaken, h_lastDesc, h_curDesc;
    int last_rounds = 0;
};

// CurrentFrame / LastFrame pose algebra of ORBmatcher.cc:1338-1349 (cv::gemm semantics: A*b+c without transposition = FP32 row
// product then FP64 alpha/beta; -A^T*b = FP64 accumulation)
static void forward_backward(const float* Tc, const float* Tl, float mb, bool mono, int& fwd, int& bwd) {
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)Tc[4 * k + r] * (double)Tc[4 * k + 3]; twc[r] = (float)(s * -1.0); }
    for (int r = 0; r < 3; r++) { const float t = Tl[4 * r] * twc[0] + Tl[4 * r + 1] * twc[1] + Tl[4 * r + 2] * twc[2]; tlc[r] = (float)((double)t * 1.0 + (double)Tl[4 * r + 3] * 1.0); }
    fwd = tlc[2] > mb && !mono; bwd = -tlc[2] > mb && !mono;
}

template <class T> static int up(DevBuf<T>& d, const std::vector<T>& h, size_t n, hipStream_t s) { HIP_TRY(hipMemcpyAsync(d.p, h.data(), n * sizeof(T), hipMemcpyHostToDevice, s)); return SIND_OK; }

extern "C" {

int sind_match_create(const sind_match_config* c, sind_match** out) {
    if (!c || !out || c->cap_last < 1 || c->cap_cur < 1 || c->max_batch < 1 || c->nlevels < 1 || c->nlevels > 16 || !(c->fx > 0)) { sind_set_error("sind_match_create: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    sind_match* m = new sind_match(); m->device = c->device; m->maxB = c->max_batch;
    sind::MatchParams& p = m->prm; p.fx = c->fx; p.fy = c->fy; p.cx = c->cx; p.cy = c->cy; p.bf = c->bf; std::memcpy(p.bounds, c->bounds, sizeof(p.bounds));
    for (int i = 0; i < 16; i++) p.scale[i] = i < c->nlevels ? c->scale_factors[i] : 0.f;
    p.nlevels = c->nlevels; p.capLast = c->cap_last; p.capCur = c->cap_cur; m->mb = c->bf / c->fx;                 // Frame.cc:167 mb = mbf / fx
    const size_t B = c->max_batch, nl = B * c->cap_last, nc = B * c->cap_cur;
    int r = SIND_OK;
    if ((r = m->pose.alloc(B)) || (r = m->nLast.alloc(B)) || (r = m->nCur.alloc(B)) || (r = m->lastOct.alloc(nl)) || (r = m->curOct.alloc(nc)) || (r = m->gstart.alloc(B * 3073)) ||
        (r = m->gidx.alloc(nc)) || (r = m->choice.alloc(nl)) || (r = m->minOwner.alloc(nc)) || (r = m->matchOfCur.alloc(nc)) || (r = m->nmatches.alloc(B)) || (r = m->rounds.alloc(B)) ||
        (r = m->x3Dw.alloc(nl * 3)) || (r = m->lastAng.alloc(nl)) || (r = m->curXY.alloc(nc * 2)) || (r = m->curAng.alloc(nc)) || (r = m->curUR.alloc(nc)) || (r = m->lastFlags.alloc(nl)) ||
        (r = m->curTaken.alloc(nc)) || (r = m->lastDesc.alloc(nl * 8)) || (r = m->curDesc.alloc(nc * 8))) { delete m; return r; }
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; sind_set_error("sind_match_create: stream creation failed"); return SIND_E_HIP; }
    m->h_pose.resize(B); m->h_nLast.resize(B); m->h_nCur.resize(B); m->h_lastOct.resize(nl); m->h_curOct.resize(nc); m->h_gstart.resize(B * 3073); m->h_gidx.resize(nc); m->h_match.resize(nc);
    m->h_nm.resize(B); m->h_rounds.resize(B); m->h_x3Dw.resize(nl * 3); m->h_lastAng.resize(nl); m->h_curXY.resize(nc * 2); m->h_curAng.resize(nc); m->h_curUR.resize(nc);
    m->h_lastFlags.resize(nl); m->h_curTaken.resize(nc); m->h_lastDesc.resize(nl * 32); m->h_curDesc.resize(nc * 32);
    *out = m; return SIND_OK;
}
int sind_match_destroy(sind_match* m) {
    if (!m) return SIND_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    hipStream_t s = m->stream; delete m; if (s) (void)hipStreamDestroy(s);
    return SIND_OK;
}

int sind_match_by_projection(sind_match* m, const sind_match_pair* pairs, int B, float th, int mono, int check_orientation) {
    if (!m || !pairs || B < 1 || B > m->maxB || !(th > 0)) { sind_set_error("sind_match_by_projection: bad arguments (B=%d, max %d)", B, m ? m->maxB : 0); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(m->device));
    sind::MatchParams p = m->prm; p.th = th; p.checkOrientation = check_orientation ? 1 : 0;
    const int cl = p.capLast, cc = p.capCur;
    for (int b = 0; b < B; b++) {
        const sind_match_pair& q = pairs[b];
        if (q.n_last < 0 || q.n_last > cl || q.n_cur < 0 || q.n_cur > cc) { sind_set_error("sind_match_by_projection: pair %d has %d / %d points, capacity %d / %d", b, q.n_last, q.n_cur, cl, cc); return SIND_E_CAPACITY; }
        if (!q.Tcw_cur || !q.Tcw_last || !q.match_of_cur || !q.nmatches || (q.n_last && (!q.x3Dw || !q.last_valid || !q.last_has_obs || !q.last_octave || !q.last_angle || !q.last_desc)) ||
            (q.n_cur && (!q.cur_un_xy || !q.cur_octave || !q.cur_angle || !q.cur_u_right || !q.cur_desc || !q.grid_idx)) || !q.grid_start) { sind_set_error("sind_match_by_projection: null array in pair %d", b); return SIND_E_ARG; }
        sind::MatchPose& ps = m->h_pose[b]; std::memcpy(ps.Tcw, q.Tcw_cur, sizeof(ps.Tcw));
        forward_backward(q.Tcw_cur, q.Tcw_last, m->mb, mono != 0, ps.forward, ps.backward);
        m->h_nLast[b] = q.n_last; m->h_nCur[b] = q.n_cur;
        for (int i = 0; i < q.n_last; i++) {
            if (q.last_octave[i] < 0 || q.last_octave[i] >= p.nlevels) { sind_set_error("sind_match_by_projection: octave %d outside [0,%d)", q.last_octave[i], p.nlevels); return SIND_E_ARG; }
            m->h_lastFlags[(size_t)b * cl + i] = (uint8_t)((q.last_valid[i] ? 1 : 0) | (q.last_has_obs[i] ? 2 : 0));
        }
        std::memcpy(&m->h_x3Dw[(size_t)b * cl * 3], q.x3Dw, (size_t)q.n_last * 12); std::memcpy(&m->h_lastOct[(size_t)b * cl], q.last_octave, (size_t)q.n_last * 4);
        std::memcpy(&m->h_lastAng[(size_t)b * cl], q.last_angle, (size_t)q.n_last * 4); std::memcpy(&m->h_lastDesc[(size_t)b * cl * 32], q.last_desc, (size_t)q.n_last * 32);
        std::memcpy(&m->h_curXY[(size_t)b * cc * 2], q.cur_un_xy, (size_t)q.n_cur * 8); std::memcpy(&m->h_curOct[(size_t)b * cc], q.cur_octave, (size_t)q.n_cur * 4);
        std::memcpy(&m->h_curAng[(size_t)b * cc], q.cur_angle, (size_t)q.n_cur * 4); std::memcpy(&m->h_curUR[(size_t)b * cc], q.cur_u_right, (size_t)q.n_cur * 4);
        std::memcpy(&m->h_curDesc[(size_t)b * cc * 32], q.cur_desc, (size_t)q.n_cur * 32);
        if (q.grid_start[0] != 0 || q.grid_start[3072] < 0 || q.grid_start[3072] > q.n_cur) { sind_set_error("sind_match_by_projection: malformed grid of pair %d", b); return SIND_E_ARG; }
        for (int c = 0; c < 3072; c++) if (q.grid_start[c + 1] < q.grid_start[c]) { sind_set_error("sind_match_by_projection: malformed grid of pair %d", b); return SIND_E_ARG; }
        for (int j = 0; j < q.grid_start[3072]; j++) if (q.grid_idx[j] < 0 || q.grid_idx[j] >= q.n_cur) { sind_set_error("sind_match_by_projection: grid index outside the keypoints (pair %d)", b); return SIND_E_ARG; }
        std::memcpy(&m->h_gstart[(size_t)b * 3073], q.grid_start, 3073 * 4); std::memcpy(&m->h_gidx[(size_t)b * cc], q.grid_idx, (size_t)q.grid_start[3072] * 4);
        if (q.cur_taken) std::memcpy(&m->h_curTaken[(size_t)b * cc], q.cur_taken, q.n_cur); else std::memset(&m->h_curTaken[(size_t)b * cc], 0, q.n_cur);
    }
    hipStream_t s = m->stream; const size_t nl = (size_t)B * cl, nc = (size_t)B * cc;
    SIND_TRY(up(m->pose, m->h_pose, B, s)); SIND_TRY(up(m->nLast, m->h_nLast, B, s)); SIND_TRY(up(m->nCur, m->h_nCur, B, s)); SIND_TRY(up(m->x3Dw, m->h_x3Dw, nl * 3, s));
    SIND_TRY(up(m->lastFlags, m->h_lastFlags, nl, s)); SIND_TRY(up(m->lastOct, m->h_lastOct, nl, s)); SIND_TRY(up(m->lastAng, m->h_lastAng, nl, s));
    HIP_TRY(hipMemcpyAsync(m->lastDesc.p, m->h_lastDesc.data(), nl * 32, hipMemcpyHostToDevice, s)); HIP_TRY(hipMemcpyAsync(m->curDesc.p, m->h_curDesc.data(), nc * 32, hipMemcpyHostToDevice, s));
    SIND_TRY(up(m->curXY, m->h_curXY, nc * 2, s)); SIND_TRY(up(m->curOct, m->h_curOct, nc, s)); SIND_TRY(up(m->curAng, m->h_curAng, nc, s)); SIND_TRY(up(m->curUR, m->h_curUR, nc, s));
    SIND_TRY(up(m->gstart, m->h_gstart, (size_t)B * 3073, s)); SIND_TRY(up(m->gidx, m->h_gidx, nc, s)); SIND_TRY(up(m->curTaken, m->h_curTaken, nc, s));
    sind::MatchArrays a{m->pose.p, m->nLast.p, m->nCur.p, m->x3Dw.p, m->lastFlags.p, m->lastOct.p, m->lastAng.p, m->lastDesc.p, m->curXY.p, m->curOct.p, m->curAng.p, m->curUR.p,
                        m->curDesc.p, m->gstart.p, m->gidx.p, m->curTaken.p, m->choice.p, m->minOwner.p, m->matchOfCur.p, m->nmatches.p, m->rounds.p};
    SIND_TRY(sind::launch_search_by_projection(p, a, B, s));
    HIP_TRY(hipMemcpyAsync(m->h_match.data(), m->matchOfCur.p, nc * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(m->h_nm.data(), m->nmatches.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(m->h_rounds.data(), m->rounds.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    m->last_rounds = 0;
    for (int b = 0; b < B; b++) {
        std::memcpy(pairs[b].match_of_cur, &m->h_match[(size_t)b * cc], (size_t)pairs[b].n_cur * 4); *pairs[b].nmatches = m->h_nm[b];
        m->last_rounds = std::max(m->last_rounds, m->h_rounds[b]);
    }
    return SIND_OK;
}
int sind_match_last_rounds(sind_match* m) { return m ? m->last_rounds : SIND_E_ARG; }

}  // extern "C"
