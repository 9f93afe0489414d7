// CPU-only test shim over the product's HOST stages (bit-image morphology, border following, homography, PEAC graph,
// octree).  Built with plain g++ into libsind_host.so so that `pytest -m "not gpu"` can exercise the host logic against
// the oracle without a GPU.  Not part of libsind_hip.so's ABI and never used by the product path.
#include <cstring>
#include "host.hpp"
#include "peac_fit4.hpp"
#include "../orb.hpp"

using namespace sind;

extern "C" {

void sindh_morph(const uint8_t* src, int w, int h, int n, int op, uint8_t* out) {
    const BitImg b = BitImg::from_u8(src, w, h, w); const EllipseElem e(n);
    const BitImg r = op == 0 ? b.dilated(e) : op == 1 ? b.eroded(e) : op == 2 ? b.opened(e) : b.closed(e);
    r.to_u8(out, w, 255);
}
// row-restricted variants used on sparse masks (ops 4 = eroded_rows, 5 = opened_rows over the mask's own row range)
void sindh_morph_rows(const uint8_t* src, int w, int h, int n, int op, uint8_t* out) {
    const BitImg b = BitImg::from_u8(src, w, h, w); const EllipseElem e(n); const Rect bb = b.bbox();
    BitImg r(w, h);
    if (!bb.empty()) r = op == 4 ? b.eroded_rows(e, bb.y0, bb.y1) : op == 5 ? b.opened_rows(e, bb.y0, bb.y1) : b.dilated(e, bb.y0, bb.y1);
    r.to_u8(out, w, 255);
}
// floodFill(FLOODFILL_MASK_ONLY) stand-in: same / blocked as byte images, returns the filled area and the filled mask
int sindh_flood_fill(const uint8_t* same, const uint8_t* blocked, int w, int h, int sx, int sy, uint8_t* filled_out, uint8_t* blocked_out) {
    const BitImg s = BitImg::from_u8(same, w, h, w); BitImg b = BitImg::from_u8(blocked, w, h, w), f(w, h);
    const int area = flood_fill(s, b, f, PtI{sx, sy});
    f.to_u8(filled_out, w, 255); b.to_u8(blocked_out, w, 255);
    return area;
}
// byte image <-> bit image round trip and per-value split (SSE2 paths)
void sindh_pack_roundtrip(const uint8_t* src, int w, int h, uint8_t v, uint8_t* nonzero_out, uint8_t* equal_out) {
    BitImg::from_u8(src, w, h, w).to_u8(nonzero_out, w, 200, 3); BitImg::from_equal(src, w, h, w, v).to_u8(equal_out, w, 255);
}
int sindh_find_contours(const uint8_t* src, int w, int h, int external_only, int* pts_xy, int cap_pts, int* lens, int cap_contours) {
    std::vector<Contour> cs; find_contours(BitImg::from_u8(src, w, h, w), cs, external_only != 0);
    int np = 0, nc = 0;
    for (auto& c : cs) { if (nc >= cap_contours) break; lens[nc++] = (int)c.size(); for (auto& p : c) { if (np < cap_pts) { pts_xy[2 * np] = p.x; pts_xy[2 * np + 1] = p.y; } np++; } }
    return (int)cs.size();
}
void sindh_draw(const uint8_t* src, int w, int h, int filled, uint8_t* out) {     // draw every external contour of src, filled or thickness 2
    const BitImg b = BitImg::from_u8(src, w, h, w); std::vector<Contour> cs; find_contours(b, cs, true);
    BitImg o(w, h);
    for (const Contour& c : cs) { if (filled) draw_filled(o, c); else draw_thick2(o, c); }
    o.to_u8(out, w, 255);
}
int sindh_find_homography(const float* src, const float* dst, int n, double* H) {
    std::vector<Pt2f> s(n), d(n); for (int i = 0; i < n; i++) { s[i] = {src[2 * i], src[2 * i + 1]}; d[i] = {dst[2 * i], dst[2 * i + 1]}; }
    return find_homography_rho(s, d, H) ? 1 : 0;
}
// block statistics in the same sequential order as k_peac_block_stats (test-only stand-in for the kernel)
void sindh_peac(const uint16_t* depth, int w, int h, float fx, float fy, float cx, float cy, float depthScale, uint8_t* out) {
    const int bw = 16, bh = 16, Nw = w / bw, Nh = h / bh;
    std::vector<PeacBlockStats> blocks((size_t)Nw * Nh);
    auto getz = [&](int i, int j, double& x, double& y, double& z) { const float d = (float)depth[(size_t)i * w + j]; if (d < 1e-3f) return false; const float zf = d * (1.0f / depthScale);
        x = (double)((j - cx) * zf / fx); y = (double)((i - cy) * zf / fy); z = (double)zf; return true; };
    for (int blk = 0; blk < Nw * Nh; blk++) {
        const int by = blk / Nw, bx = blk - by * Nw;
        PeacBlockStats S; std::memset(&S, 0, sizeof(S)); S.valid = 1;
        for (int i = by * bh; i < (by + 1) * bh && i < h && S.valid; i++)
            for (int j = bx * bw; j < (bx + 1) * bw && j < w; j++) {
                double x, y, z, xn, yn, zn;
                if (!getz(i, j, x, y, z)) { S.valid = 0; break; }
                if (j + 1 < w && getz(i, j + 1, xn, yn, zn) && std::fabs(z - zn) > 0.04 * std::fabs(z) + 20.0) { S.valid = 0; break; }
                if (i + 1 < h && getz(i + 1, j, xn, yn, zn) && std::fabs(z - zn) > 0.04 * std::fabs(z) + 20.0) { S.valid = 0; break; }
                S.sx += x; S.sy += y; S.sz += z; S.sxx += x * x; S.syy += y * y; S.szz += z * z; S.sxy += x * y; S.syz += y * z; S.sxz += x * z; S.N++;
            }
        blocks[blk] = S;
    }
    BitImg pc; PeacInput in{blocks.data(), depth, w, h, fx, fy, cx, cy, depthScale};
    peac_plane_contours(in, pc);
    pc.to_u8(out, w, 255);
}
// BitImg::dilation_hits against dilated().get at n query points (test of the CalOccluded contour filter's shortcut): returns the number of disagreements
int sindh_dilation_hits_check(const uint8_t* src, int w, int h, int elem, const int* xy, int n) {
    const BitImg b = BitImg::from_u8(src, w, h, w); const EllipseElem e(elem); const BitImg d = b.dilated(e);
    int bad = 0; for (int i = 0; i < n; i++) bad += d.get(xy[2 * i], xy[2 * i + 1]) != b.dilation_hits(e, xy[2 * i], xy[2 * i + 1]);
    return bad;
}
// n plane fits from their moments (9 doubles each) and point counts, four at a time (peac_fit4) or one at a time (peac_fit): out = n x {centre 3, normal 3, mse}
void sindh_peac_fits(const double* moments, const int* counts, int n, int four_lanes, double* out) {
    for (int i = 0; i < n; i += 4) {
        const int c = std::min(4, n - i); PeacFitIn in[4]; PeacFitOut o[4];
        for (int k = 0; k < c; k++) { std::memcpy(in[k].m, moments + 9 * (size_t)(i + k), 72); in[k].N = counts[i + k]; }
        if (four_lanes) peac_fit4(in, c, o);
        else for (int k = 0; k < c; k++) peac_fit(in[k].m, in[k].N, o[k].center, o[k].normal, o[k].mse);
        for (int k = 0; k < c; k++) { double* r = out + 7 * (size_t)(i + k); for (int q = 0; q < 3; q++) { r[q] = o[k].center[q]; r[3 + q] = o[k].normal[q]; } r[6] = o[k].mse; }
    }
}
int sindh_octree(const float* xyr, int n, int minX, int maxX, int minY, int maxY, int N, float* out_xyr, int cap) {
    std::vector<OctKp> in(n), out; for (int i = 0; i < n; i++) in[i] = {xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
    distribute_octree(in, minX, maxX, minY, maxY, N, out);
    for (int i = 0; i < (int)out.size() && i < cap; i++) { out_xyr[3 * i] = out[i].x; out_xyr[3 * i + 1] = out[i].y; out_xyr[3 * i + 2] = out[i].response; }
    return (int)out.size();
}

}  // extern "C"
