// Host-side geometry: everything the reference derives from (constructor args, image size) before it
// touches a pixel, computed once per (w,h) and uploaded to the device as tables.
//   constructor tables      reference src/ORBextractor.cc:457-511
//   level sizes             reference src/ORBextractor.cc:783-786 (sizes from the ORIGINAL image)
//   cell grid               reference src/ORBextractor.cc:527-596
//   cv::resize coefficients OpenCV 2.4 imgwarp.cpp
// The float/double mix of every expression is kept exactly as in the reference, since the integer
// results (level sizes, quotas, grid shape) depend on it.
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "orbx_internal.h"

namespace orbx {

static inline int round_even(double v) { return (int)std::lrint(v); }   // cvRound
static inline int floor_int(double v) { int i = (int)v; return i - (i > v); }   // cvFloor
static inline int ceil_int(double v) { int i = (int)v; return i + (i < v); }    // cvCeil
static inline int16_t sat16(float v) {   // saturate_cast<short>(float)
    int iv = round_even(v);
    return (int16_t)(iv < SHRT_MIN ? SHRT_MIN : iv > SHRT_MAX ? SHRT_MAX : iv);
}
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int build_geometry_band(const orbx_params& p, int w, int h, HostGeom& out, std::string& err, const int band_px, const FastShape& shape) {
    const int threads = shape.threads;
    const int nl = p.nlevels;
    if (nl < 1 || nl > MAX_LEVELS) { err = "nlevels out of range [1,16]"; return ORBX_ERR_ARG; }
    if (p.nfeatures < 1) { err = "nfeatures must be >= 1"; return ORBX_ERR_ARG; }
    if (!(p.scale_factor > 1.0f) || p.scale_factor > 2.5f) { err = "scale_factor must be in (1, 2.5]"; return ORBX_ERR_ARG; }
    if (w < 1 || h < 1 || w > 16384 || h > 16384) { err = "image size out of range"; return ORBX_ERR_ARG; }
    if (p.score_type != ORBX_HARRIS_SCORE && p.score_type != ORBX_FAST_SCORE) { err = "score_type"; return ORBX_ERR_ARG; }

    out = HostGeom();
    DevGeom& g = out.g;
    memset(&g, 0, sizeof(g));
    g.nlevels = nl;
    g.score_type = p.score_type;
    g.fp_contract = p.fp_contract ? 1 : 0;
    g.fast_th = std::min(std::max(p.fast_th, 0), 255);   // cv::FAST clamps its threshold
    g.tmin = std::min(g.fast_th, 7);                    // one score pass serves fastTh and the fallback 7

    // --- constructor tables (:457-487).  scaleFactor is a double member initialised from a float.
    const double scaleFactor = (double)p.scale_factor;
    out.scale.assign(nl, 1.f);
    out.inv_scale.assign(nl, 1.f);
    for (int i = 1; i < nl; i++) out.scale[i] = (float)(out.scale[i - 1] * scaleFactor);
    const float invScaleFactor = (float)(1.0f / scaleFactor);
    for (int i = 1; i < nl; i++) out.inv_scale[i] = out.inv_scale[i - 1] * invScaleFactor;
    out.features_per_level.assign(nl, 0);
    {
        const float factor = (float)(1.0 / scaleFactor);
        float per_scale = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) {
            out.features_per_level[l] = round_even(per_scale);
            sum += out.features_per_level[l];
            per_scale *= factor;
        }
        out.features_per_level[nl - 1] = std::max(p.nfeatures - sum, 0);
    }
    // circular-patch row extents (:495-510)
    {
        int* umax = g.umax;
        int v, v0;
        const int vmax = floor_int(HALF_PATCH * std::sqrt(2.f) / 2 + 1);
        const int vmin = ceil_int(HALF_PATCH * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH * HALF_PATCH;
        for (v = 0; v <= vmax; ++v) umax[v] = round_even(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        // the describe kernel carries this table as a packed constant
        const unsigned long long nib = 0x3689ABCDDEEEFFFFull;
        for (v = 0; v <= HALF_PATCH; ++v)
            if (umax[v] != (int)((nib >> (4 * v)) & 15ull)) { err = "umax table mismatch"; return ORBX_ERR_ARG; }
        // ... and relies on the patch being symmetric: |u| <= umax[|v|]  <=>  |v| <= umax[|u|]
        for (int a = 0; a <= HALF_PATCH; ++a)
            for (int c = 0; c <= HALF_PATCH; ++c)
                if ((a <= umax[c]) != (c <= umax[a])) { err = "circular patch not symmetric"; return ORBX_ERR_ARG; }
    }

    // --- per-level geometry
    const float imageRatio = (float)w / h;   // level 0 cols/rows (:526)
    int plane_off = 0, cell_base = 0, cand_base = 0, sel_base = 0, slot_base = 0, quad_base = 0;
    int btile_base = 0, btile_base_s = 0, mb_base = 0;
    for (int l = 0; l < nl; l++) {
        LevelGeom& L = g.lv[l];
        const float s = out.inv_scale[l];
        L.w = round_even((float)w * s);
        L.h = round_even((float)h * s);
        if (L.w < 2 * EDGE + 7 || L.h < 2 * EDGE + 7) {
            // the reference would build cell views with negative extent here (cv::Exception)
            err = "level " + std::to_string(l) + " too small for the 16-px border + FAST ring";
            return ORBX_ERR_GEOMETRY;
        }
        L.stride = align_up(L.w, 64);
        L.plane_off = plane_off;
        plane_off += L.stride * L.h;
        L.scale = out.scale[l];
        L.kp_size = (float)(int)(31 * out.scale[l]);
        L.ndesired = out.features_per_level[l];
        L.blur_wvec = (p.blur_rounding == ORBX_BLUR_X86_SSE2) ? (L.w & ~3) : 0;
        {
            // k_blur border lanes: v_perm_b32(S0 = D_last (px q..q+3), S1 = D_prev (px q-4..q-1)), selector byte = source index
            // 0..3 -> D_prev, 4..7 -> D_last.  px p lives at index p-(q-4); out-of-row px p >= w reflect to 2w-2-p.
            const int q = (L.w - 1) & ~3, rem = L.w - q;
            unsigned sel_last = 0, sel_halo = 0;
            for (int i = 0; i < 4; i++) {
                const int idx_last = i < rem ? 4 + i : 2 * rem + 2 - i;          // px q+i
                int idx_halo = 2 * rem - 2 - i;                                    // px q+4+i (only px <= w+2 are ever used)
                if (idx_halo < 0) idx_halo = 0;
                sel_last |= (unsigned)(idx_last & 7) << (8 * i);
                sel_halo |= (unsigned)(idx_halo & 7) << (8 * i);
            }
            L.blur_sel_last = (int)sel_last;
            L.blur_sel_halo = (int)sel_halo;
        }

        // grid (:534-547)
        const int levelCols = (int)std::sqrt((float)L.ndesired / (5 * imageRatio));
        const int levelRows = (int)(imageRatio * levelCols);
        if (levelCols < 1 || levelRows < 1) {
            err = "level " + std::to_string(l) + ": empty cell grid (the reference divides by zero here)";
            return ORBX_ERR_GEOMETRY;
        }
        const int minB = EDGE, maxBX = L.w - EDGE, maxBY = L.h - EDGE;
        const int W = maxBX - minB, H = maxBY - minB;
        L.gcols = levelCols;
        L.grows = levelRows;
        L.cellW = (int)std::ceil((float)W / levelCols);
        L.cellH = (int)std::ceil((float)H / levelRows);
        L.ncells = levelRows * levelCols;
        L.nfeat_cell = (int)std::ceil((float)L.ndesired / L.ncells);
        if (L.ncells > QUOTA_MAX_CELLS) { err = "more than " + std::to_string(QUOTA_MAX_CELLS) + " grid cells on a level"; return ORBX_ERR_CAPACITY; }   // k_quota keeps a level's cells in LDS
        g.quota_cells = std::max(g.quota_cells, align_up(L.ncells, 64));
        // every cell but the last of a row/column keeps its full cellW+6 view: it must fit the level
        if ((levelCols - 1) * L.cellW > W || (levelRows - 1) * L.cellH > H) {
            err = "level " + std::to_string(l) + ": degenerate cell grid (cell views leave the image in the reference)";
            return ORBX_ERR_GEOMETRY;
        }
        L.cell_base = cell_base;
        L.cand_base = cand_base;
        int cand_off = 0;
        for (int i = 0; i < levelRows; i++)
            for (int j = 0; j < levelCols; j++) {
                CellGeom c;
                c.x0 = (int16_t)(minB + j * L.cellW);
                c.y0 = (int16_t)(minB + i * L.cellH);
                c.x1 = (int16_t)((j == levelCols - 1) ? maxBX - 1 : c.x0 + L.cellW - 1);
                c.y1 = (int16_t)((i == levelRows - 1) ? maxBY - 1 : c.y0 + L.cellH - 1);
                // reference: hX = maxBorderX+3-iniX <= 0  <=>  x0 >= maxBorderX+6  (iniX = x0-3)
                const bool skipX = (j == levelCols - 1) && (maxBX + 3 - (c.x0 - 3) <= 0);
                const bool skipY = (i == levelRows - 1) && (maxBY + 3 - (c.y0 - 3) <= 0);
                c.skipped = (skipX || skipY) ? 1 : 0;
                const int cw = std::max(0, c.x1 - c.x0 + 1), ch = std::max(0, c.y1 - c.y0 + 1);
                c.cand_off = cand_off;
                c.band0 = (int)out.bands.size();
                const int nb = std::max(1, (cw * ch + band_px - 1) / band_px);
                const int rb = std::max(1, (ch + nb - 1) / nb);
                c.nbands = 0;
                c.cand_cap = 0;
                for (int k = 0; k < nb; k++) {
                    BandGeom bg;
                    memset(&bg, 0, sizeof(bg));
                    bg.x0 = c.x0; bg.x1 = c.x1;
                    bg.y0 = (int16_t)(c.y0 + k * rb);
                    bg.y1 = (int16_t)std::min<int>(c.y1, bg.y0 + rb - 1);
                    if (k > 0 && bg.y0 > c.y1) break;                 // (rounding left no rows for this band)
                    bg.ey0 = (int16_t)std::max<int>(c.y0, bg.y0 - 1);
                    bg.ey1 = (int16_t)std::min<int>(c.y1, bg.y1 + 1);
                    bg.level = l;
                    const int bh = std::max(0, bg.y1 - bg.y0 + 1);
                    bg.cand_off = cand_off;
                    bg.cand_cap = ((cw + 1) / 2) * ((bh + 1) / 2);    // strict 3x3 maxima cannot be adjacent
                    if (cw > 0) {                                     // (same row geometry as fast_band_task derives from x0 and cw)
                        const int nd = fast_row_dwords((bg.x0 - 3) & 3, cw);
                        bg.inv_nd = 1.0f / (float)nd;
                        bg.inv_s = 1.0f / (float)(nd * 4);
                        bg.inv_cpr = 1.0f / (float)(nd >> 2);
                    }
                    cand_off += bg.cand_cap;
                    c.cand_cap += bg.cand_cap;
                    c.nbands++;
                    out.bands.push_back(bg);
                }
                out.cells.push_back(c);
            }
        cell_base += L.ncells;
        cand_base += cand_off;
        L.sel_base = sel_base;
        L.sel_cap = std::min(cand_off, L.ncells * L.nfeat_cell + L.ncells * L.ncells + L.ndesired + 64);
        sel_base += L.sel_cap;
        L.slot_base = slot_base;
        slot_base += L.ndesired;
        L.quad_base = quad_base;
        quad_base += (L.ndesired + 3) / 4;

        // blur work items
        L.btiles_x = (L.w + 247) / 248;        // blur: 248-px column strips x 32-row bands, one wave each
        L.btiles_y = (L.h + BLUR_ROWS - 1) / BLUR_ROWS;
        L.btile_base = btile_base;
        btile_base += L.btiles_x * L.btiles_y;
        L.btiles_y_s = (L.h + BLUR_ROWS_SMALL - 1) / BLUR_ROWS_SMALL;
        L.btile_base_s = btile_base_s;
        btile_base_s += L.btiles_x * L.btiles_y_s;
        {
            const int nsteps = (L.h + 31) / 32;
            L.mb_strips = (L.w + 32 * MB_TILES - 1) / (32 * MB_TILES);
            L.mb_bands = (nsteps + MB_BAND_STEPS - 1) / MB_BAND_STEPS;
            L.mb_band_steps = (nsteps + L.mb_bands - 1) / L.mb_bands;
            L.mb_n = L.mb_strips * L.mb_bands;
        }
        L.mb_base = mb_base;
        mb_base += L.mb_n;

        // cv::resize tables level l-1 -> l
        if (l > 0) {
            const int sw = g.lv[l - 1].w, sh = g.lv[l - 1].h, dw = L.w, dh = L.h;
            if (sw > 32767 || sh > 32767) { err = "image too large"; return ORBX_ERR_ARG; }
            const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
            const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
            L.tabx_off = (int)out.tabx.size();
            L.taby_off = (int)out.taby.size();
            for (int dx = 0; dx < dw; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = floor_int(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
                ResizeX e;
                e.sx = (int16_t)sx;
                e.sx1 = (int16_t)std::min(sx + 1, sw - 1);
                e.a0 = sat16((1.f - fx) * 2048);
                e.a1 = sat16(fx * 2048);
                out.tabx.push_back(e);
            }
            for (int dy = 0; dy < dh; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = floor_int(fy);
                fy -= sy;
                ResizeY e;
                e.sy0 = (int16_t)std::min(std::max(sy, 0), sh - 1);
                e.sy1 = (int16_t)std::min(std::max(sy + 1, 0), sh - 1);
                e.b0 = sat16((1.f - fy) * 2048);
                e.b1 = sat16(fy * 2048);
                out.taby.push_back(e);
            }
            // LDS tile of k_resize: the source rectangle of every 256 x 16 output tile (tables are monotone)
            {
                const ResizeX* tx = out.tabx.data() + L.tabx_off;
                const ResizeY* ty = out.taby.data() + L.taby_off;
                int nd_max = 1, nr_max = 1;
                for (int bx0 = 0; bx0 < dw; bx0 += 256) {
                    const int bx1 = std::min(bx0 + 255, dw - 1);
                    nd_max = std::max(nd_max, ((tx[bx1].sx1 - (tx[bx0].sx & ~3)) >> 2) + 1);
                }
                for (int by0 = 0; by0 < dh; by0 += RZ_ROWS) {
                    const int by1 = std::min(by0 + RZ_ROWS - 1, dh - 1);
                    nr_max = std::max(nr_max, ty[by1].sy1 - ty[by0].sy0 + 1);
                }
                L.rz_pitch = 4 * nd_max + 8;      // + 8: the windowed form reads three dwords from the aligned start of a lane's source window
                L.rz_rows = nr_max;
                // widest source window of four adjacent output pixels (a lane of k_resize): up to 8 bytes it is cut out of three
                // aligned LDS dwords with v_alignbyte and addressed by per-lane v_perm selectors
                int span = 1;
                for (int x0 = 0; x0 < dw; x0 += 4) span = std::max(span, tx[std::min(x0 + 3, dw - 1)].sx1 - tx[x0].sx + 1);
                L.rz_window = span <= 8 ? 1 : 0;
                if (L.rz_pitch * L.rz_rows > 64 * 1024) { err = "resize tile does not fit the LDS"; return ORBX_ERR_CAPACITY; }
            }
        }
    }
    // fused pyramid launches: groups of levels, per-tile regions of every level of a group (see PyrGroup)
    {
        constexpr int tile[2] = {ORBX_PYR_TILE};
        g.npyr_groups = 0;
        out.pyr_tab.clear();
        bool ok = scaleFactor <= 2.0 && nl > 1 && !getenv("ORBX_PYR_PER_LEVEL");      // beyond 2 the bilinear taps of neighbouring pixels leave gaps
        int l0 = 0;
        while (ok && l0 < nl - 1) {
            if (g.npyr_groups == PYR_MAX_GROUPS) { ok = false; break; }
            PyrGroup pg;
            memset(&pg, 0, sizeof(pg));
            pg.l0 = l0;
            constexpr int depth_cfg[2] = {ORBX_PYR_DEPTH};   // first group, further groups
            pg.depth = std::min(std::min(l0 == 0 ? depth_cfg[0] : depth_cfg[1], PYR_MAX_DEPTH), nl - 1 - l0);
            const LevelGeom& LD = g.lv[l0 + pg.depth];
            pg.ntx = (LD.w + tile[0] - 1) / tile[0];
            pg.nty = (LD.h + tile[1] - 1) / tile[1];
            // regions[k][i] = {first, last} column of tile i on level l0 + k; entry ntx is the end sentinel {W, W}
            auto chain = [&](int n, int tsz, bool is_x, std::vector<int>& tab) {
                const int base = (int)tab.size();
                tab.resize(base + (size_t)(pg.depth + 1) * (n + 1) * 2);
                auto at = [&](int k, int i, int e) -> int& { return tab[base + ((size_t)k * (n + 1) + i) * 2 + e]; };
                const int dim_d = is_x ? LD.w : LD.h;
                for (int i = 0; i <= n; i++) { at(pg.depth, i, 0) = std::min(i * tsz, dim_d); at(pg.depth, i, 1) = std::min((i + 1) * tsz, dim_d) - 1; }
                at(pg.depth, n, 1) = dim_d;
                for (int k = pg.depth; k >= 1; k--) {
                    const LevelGeom& Lk = g.lv[l0 + k];
                    const LevelGeom& Ls = g.lv[l0 + k - 1];
                    const int dk = is_x ? Lk.w : Lk.h, ds = is_x ? Ls.w : Ls.h;
                    // starts first (they define ownership), then the ends: what the region's last pixel needs, at least up to the next start
                    for (int i = 0; i <= n; i++) {
                        const int s0 = at(k, i, 0);
                        int src = ds;
                        if (s0 < dk) src = is_x ? (out.tabx[Lk.tabx_off + s0].sx & ~3) : out.taby[Lk.taby_off + s0].sy0;
                        at(k - 1, i, 0) = src;
                    }
                    for (int i = 0; i < n; i++) {
                        const int e0 = std::min(at(k, i, 1), dk - 1);
                        int need = is_x ? out.tabx[Lk.tabx_off + e0].sx1 : out.taby[Lk.taby_off + e0].sy1;
                        need = std::max(need, at(k - 1, i + 1, 0) - 1);
                        if (is_x) need = std::min(need | 3, ds - 1);
                        at(k - 1, i, 1) = std::max(need, at(k - 1, i, 0));
                    }
                    at(k - 1, n, 1) = ds;
                }
                return base;
            };
            pg.xtab = chain(pg.ntx, tile[0], true, out.pyr_tab);
            pg.ytab = chain(pg.nty, tile[1], false, out.pyr_tab);
            int off = 0;
            for (int k = 0; k < pg.depth; k++) {
                int maxw = 4, maxh = 1;
                for (int i = 0; i < pg.ntx; i++) maxw = std::max(maxw, out.pyr_tab[pg.xtab + ((size_t)k * (pg.ntx + 1) + i) * 2 + 1] - out.pyr_tab[pg.xtab + ((size_t)k * (pg.ntx + 1) + i) * 2] + 1);
                for (int i = 0; i < pg.nty; i++) maxh = std::max(maxh, out.pyr_tab[pg.ytab + ((size_t)k * (pg.nty + 1) + i) * 2 + 1] - out.pyr_tab[pg.ytab + ((size_t)k * (pg.nty + 1) + i) * 2] + 1);
                pg.pitch[k] = align_up(maxw, 4) + 4;
                pg.lds_off[k] = off;
                off += align_up(pg.pitch[k] * maxh, 16);
            }
            pg.lds_bytes = off;
            if (off > 64 * 1024) { ok = false; break; }
            g.pyr[g.npyr_groups++] = pg;
            l0 += pg.depth;
        }
        if (!ok) { g.npyr_groups = 0; out.pyr_tab.clear(); }
    }
    // LDS carve of k_fast_cells, sized by the largest band
    {
        int max_px = 0, max_img = 0, max_chunks = 0;
        for (const BandGeom& c : out.bands) {
            const int cw = c.x1 - c.x0 + 1, ch = c.ey1 - c.ey0 + 1;
            if (cw <= 0 || ch <= 0) continue;
            max_px = std::max(max_px, cw * ch);
            if (cw > shape.max_cw) { err = "grid cell wider than " + std::to_string(shape.max_cw) + " pixels"; return ORBX_ERR_CAPACITY; }   // two own rows + halo of the staged band must fit 64 KiB (16-bit pixel offsets)
            const int nd = fast_row_dwords(3, cw);
            max_img = std::max(max_img, nd * 4 * (ch + 6));
            max_chunks = std::max(max_chunks, (nd * 4 * (c.y1 - c.y0 + 1) + 63) / 64);
        }
        // pixel offsets inside the staged band are 16-bit, dword indices 14-bit
        if (max_px > 65535 || max_img > 65536) { err = "grid cell band larger than 64 KiB with its halo"; return ORBX_ERR_CAPACITY; }
        g.fast_max_img = align_up(std::max(max_img, 16), 16);
        g.fast_max_chunks = align_up(max_chunks + 1, 2);
        g.fast_lds_bytes = 64 /*sizeof(FastHdr)*/ + g.fast_max_chunks * 8 + (threads / 64) * fast_wave_queue_bytes(shape.ppt) + 2 * g.fast_max_img + 16;
        if (g.fast_lds_bytes > 160 * 1024) { err = "grid cell does not fit the 160 KiB LDS"; return ORBX_ERR_CAPACITY; }
    }
    {
        int max_cell = 1, max_level = 1;
        for (const CellGeom& c : out.cells) max_cell = std::max(max_cell, (int)c.cand_cap);
        for (int l = 0; l < nl; l++) max_level = std::max(max_level, g.lv[l].sel_cap);
        // lists are staged in LDS up to 2048 entries (8 B + two uint16 scratch words each); longer ones (possible only in
        // pathological images) take a sequential global-memory path
        g.sel_lds_entries = std::min(std::max(max_cell, max_level), 2048);
        g.sel_lds_cell = align_up(g.sel_lds_entries * ((int)sizeof(Cand) + 4), 16);
        g.sel_lds_level = g.sel_lds_cell;
        if (g.sel_lds_cell > 160 * 1024 || g.sel_lds_level > 160 * 1024) { err = "keypoint list does not fit the 160 KiB LDS"; return ORBX_ERR_CAPACITY; }
    }
    for (int l = 0; l < MAX_LEVELS; l++) {
        const bool live = l < nl;
        g.cell_bases[l] = live ? g.lv[l].cell_base : INT_MAX;
        g.quad_bases[l] = live ? g.lv[l].quad_base : INT_MAX;
        g.btile_bases[l] = live ? g.lv[l].btile_base : INT_MAX;
        g.btile_bases_s[l] = live ? g.lv[l].btile_base_s : INT_MAX;
        g.mb_bases[l] = live ? g.lv[l].mb_base : INT_MAX;
    }
    g.ncells_total = cell_base;
    g.nbands_total = (int)out.bands.size();
    g.nbands_magic = g.nbands_total > 1 ? (uint32_t)((1ull << 32) / (unsigned)g.nbands_total) + 1u : 0u;
    g.nbtiles_total = btile_base;
    g.nbtiles_total_s = btile_base_s;
    g.nmb_total = mb_base;
    g.nslots = slot_base;
    g.nquads = quad_base;
    g.frame_plane_bytes = align_up(plane_off, 256);
    g.frame_cands = cand_base;
    g.frame_sel = sel_base;
    return ORBX_OK;
}


// Shape of k_fast_cells (orbx_internal.h: FAST_SMALL / FAST_LARGE).  VGA-class grids (largest cell view <= 12288 px and at most
// 500 px wide) take the small shape: 256 threads over 8192-pixel bands.  Everything else takes the large shape: since round 3
// 256 threads over 7168-pixel bands (rounds 1-2: 512 threads over 10240-pixel bands); grids whose widest cell pushes the LDS
// footprint of a work item above a quarter of the CU's LDS get smaller bands from the loop below (down to 4096 px).
int build_geometry(const orbx_params& p, int w, int h, HostGeom& out, std::string& err) {
    constexpr int LDS_FOR_FOUR = (160 * 1024) / 4 - 64;
    const FastShape small = FAST_SMALL, large = FAST_LARGE;
    {
        HostGeom trial;
        std::string e2;
        if (build_geometry_band(p, w, h, trial, e2, small.band_px, small) == ORBX_OK) {
            int max_cell_px = 0;
            for (const CellGeom& c : trial.cells) max_cell_px = std::max(max_cell_px, (c.x1 - c.x0 + 1) * (c.y1 - c.y0 + 1));
            if (max_cell_px <= 12288) {
                out = trial;
                out.g.fast_threads = small.threads;
                out.g.fast_small = 1;
                err.clear();
                return ORBX_OK;
            }
        }
    }
    int rc = build_geometry_band(p, w, h, out, err, large.band_px, large);
    if (rc != ORBX_OK) return rc;
    out.g.fast_threads = large.threads;
    out.g.fast_small = 0;
    if (out.g.fast_lds_bytes <= LDS_FOR_FOUR) return rc;
    for (int band = large.band_px - 512; band >= 4096; band -= 512) {      // (re-based in round 4: the large shape's bands start at 7168 px)
        HostGeom trial;
        std::string e2;
        if (build_geometry_band(p, w, h, trial, e2, band, large) == ORBX_OK && trial.g.fast_lds_bytes <= LDS_FOR_FOUR) {
            out = trial;
            out.g.fast_threads = large.threads;
            out.g.fast_small = 0;
            err.clear();
            return ORBX_OK;
        }
    }
    return rc;
}

}  // namespace orbx
