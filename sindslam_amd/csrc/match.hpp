// Projection matcher interface (reference src/ORBmatcher.cc:1328-1470); kernel in match_kernels.hip.
#pragma once
#include "common.hpp"

namespace sind {

struct MatchParams {
    float fx, fy, cx, cy, bf, bounds[4], th; float scale[16]; int nlevels;
    int capLast, capCur, checkOrientation;
};
struct MatchPose { float Tcw[12]; int forward, backward; };       // per pair: rows 0..2 of CurrentFrame.mTcw, bForward / bBackward

struct MatchArrays {                                              // device pointers, dense [B][cap...]
    const MatchPose* pose; const int* nLast; const int* nCur;
    const float* x3Dw; const uint8_t* lastFlags /* bit0 valid, bit1 has observations */; const int* lastOctave; const float* lastAngle; const uint32_t* lastDesc;
    const float* curUnXY; const int* curOctave; const float* curAngle; const float* curURight; const uint32_t* curDesc; const int* gridStart; const int* gridIdx;
    const uint8_t* curTaken;
    int* choice; int* minOwner;                                   // scratch [B][capLast], [B][capCur]
    int* matchOfCur; int* nmatches; int* rounds;                  // outputs [B][capCur], [B], [B]
};

int launch_search_by_projection(const MatchParams& p, const MatchArrays& a, int B, hipStream_t s);

}  // namespace sind
