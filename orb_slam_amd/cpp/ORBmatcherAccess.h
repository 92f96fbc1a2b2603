// What ORBmatcher.cc needs from ORB_SLAM's Frame / KeyFrame in BULK where the reference's classes only serve it per query:
// the 64 x 48 search grid of a frame as one CSR (the device searches stage a whole frame's grid at once instead of calling
// GetFeaturesInArea per map point) and the undistorted image bounds its windows are computed with.
//
// This file is the version for ORB_SLAM's own classes (reference include/Frame.h, include/KeyFrame.h).  A build against other
// Frame / KeyFrame types names its own version with -DORBMATCHER_ACCESS_HEADER='"..."' (the test build here does:
// oracle/matcherstub/access.h, for the plain-data stand-ins the reference's ORBmatcher.cc is pinned with).
//
//   Frame     mGrid is a public member (include/Frame.h:90): flattened as it is.
//   KeyFrame  mGrid, mvKeysUn and the bounds are protected (include/KeyFrame.h:172-197) and there is no bulk accessor; the grid is
//             therefore rebuilt from public data by the rule that filled it: KeyFrame copies Frame::mGrid (src/KeyFrame.cc:43-50),
//             which Frame::Frame fills in key point order with Frame::PosInGrid (src/Frame.cc:108-123, :267-277); the bounds and the
//             inverse cell sizes are the camera's, computed once (Frame's statics; src/KeyFrame.cc:31-34 copies them).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "orbf.h"

namespace ORB_SLAM {
namespace orbm_access {

inline orbf_bounds CameraBounds() {
    orbf_bounds b;
    b.min_x = Frame::mnMinX; b.max_x = Frame::mnMaxX; b.min_y = Frame::mnMinY; b.max_y = Frame::mnMaxY;
    b.inv_w = Frame::mfGridElementWidthInv; b.inv_h = Frame::mfGridElementHeightInv;
    return b;
}

inline void GridOf(const Frame& F, orbf_bounds& b, std::vector<int32_t>& cell_off, std::vector<int32_t>& cell_feat) {
    b = CameraBounds();
    cell_off.assign(ORBF_GRID_CELLS + 1, 0);
    cell_feat.clear();
    for (int x = 0; x < ORBF_GRID_COLS; x++)
        for (int y = 0; y < ORBF_GRID_ROWS; y++) {
            const std::vector<std::size_t>& cell = F.mGrid[x][y];
            for (std::size_t j = 0; j < cell.size(); j++) cell_feat.push_back((int32_t)cell[j]);
            cell_off[x * ORBF_GRID_ROWS + y + 1] = (int32_t)cell_feat.size();
        }
}

inline void GridOf(KeyFrame* pKF, orbf_bounds& b, std::vector<int32_t>& cell_off, std::vector<int32_t>& cell_feat) {
    b = CameraBounds();
    b.inv_w = pKF->mfGridElementWidthInv; b.inv_h = pKF->mfGridElementHeightInv;
    const std::vector<cv::KeyPoint> keys = pKF->GetKeyPointsUn();
    std::vector<int32_t> cell_of(keys.size(), -1);
    cell_off.assign(ORBF_GRID_CELLS + 1, 0);
    for (std::size_t i = 0; i < keys.size(); i++) {
        const int px = (int)std::round((keys[i].pt.x - b.min_x) * b.inv_w), py = (int)std::round((keys[i].pt.y - b.min_y) * b.inv_h);
        if (px < 0 || px >= ORBF_GRID_COLS || py < 0 || py >= ORBF_GRID_ROWS) continue;      // left the image when undistorted: in no cell
        cell_of[i] = px * ORBF_GRID_ROWS + py;
        cell_off[cell_of[i] + 1]++;
    }
    for (int c = 0; c < ORBF_GRID_CELLS; c++) cell_off[c + 1] += cell_off[c];
    cell_feat.assign(cell_off[ORBF_GRID_CELLS], 0);
    std::vector<int32_t> fill(cell_off.begin(), cell_off.end() - 1);
    for (std::size_t i = 0; i < keys.size(); i++)
        if (cell_of[i] >= 0) cell_feat[fill[cell_of[i]]++] = (int32_t)i;
}

// pKF2's mvLevelSigma2 (CheckDistEpipolarLine reads it through GetSigma2)
inline std::vector<float> LevelSigma2Of(KeyFrame* pKF) { return pKF->GetVectorScaleSigma2(); }

}  // namespace orbm_access
}  // namespace ORB_SLAM
