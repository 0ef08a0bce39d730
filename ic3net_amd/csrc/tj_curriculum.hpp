// tj_curriculum.hpp — host-side scalars of the Traffic-Junction curriculum (traffic_junction_env.py:196-200 gating +
// :620-626): every env of a handle sees the same epoch sequence, so they share one add_rate exactly like N reference
// instances.  Plain C++ (no HIP): used by ic3_env_reset and by the host build of the device functions (tests/host).
#pragma once
#include <cmath>

namespace ic3 {

// Python's float `//` (CPython float_floor_div): quirk Q16 of the reference's `0.01 * (exact_rate // 0.01)`
inline double py_float_floordiv(double vx, double wx)
{
    double mod = std::fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0 && ((wx < 0) != (mod < 0))) div -= 1.0;
    if (div == 0.0) return std::copysign(0.0, vx / wx);
    double fl = std::floor(div);
    if (div - fl > 0.5) fl += 1.0;
    return fl;
}

// reset(epoch): epoch < 0 = no epoch given
inline void tj_curriculum_update(double add_rate_min, double add_rate_max, double curr_start, double curr_end, int epoch,
                                 double& exact_rate, double& add_rate, double& epoch_last_update)
{
    const double epoch_range = curr_end - curr_start, rate_range = add_rate_max - add_rate_min;
    if (epoch >= 0 && epoch_range > 0 && rate_range > 0 && (double)epoch > epoch_last_update) {
        if (curr_start <= (double)epoch && (double)epoch < curr_end) {
            const double step = rate_range / epoch_range;
            exact_rate = exact_rate + step;
            add_rate = 0.01 * py_float_floordiv(exact_rate, 0.01);
        }
        epoch_last_update = (double)epoch;
    }
}

// u <= add_rate  <=>  x24 <= floor(add_rate * 2^24)   (exact: power-of-two scaling in fp64; TJ:375)
inline int tj_rate_threshold(double add_rate)
{
    double thr_d = std::floor(add_rate * 16777216.0);
    if (thr_d > 16777216.0) thr_d = 16777216.0;
    if (thr_d < -1.0) thr_d = -1.0;
    return (int)thr_d;
}

}  // namespace ic3
