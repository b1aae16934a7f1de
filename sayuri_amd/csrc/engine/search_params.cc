#include "search_params.h"

#include <cmath>
#include <stdexcept>

namespace sayuri_engine {

namespace {
constexpr float kTwoOverPi = 0.63661977236758134308f;
inline float SmoothScore(float score, float center, float scale, float board_size) {
    return std::atan((score - center) / (scale * board_size)) * kTwoOverPi;
}
} // namespace

ScoreUtility::ScoreUtility() : table_(static_cast<size_t>(kMeanLen) * kStddevLen) {
    // numeric integration over +-5 sigma in 1/10 steps, on a 19x19-normalised score axis
    const int steps = 10, sigmas = 5;
    const int half = sigmas * steps;
    std::vector<float> pdf(2 * half + 1);
    for (int i = -half; i <= half; ++i) {
        const float x = static_cast<float>(i) / steps;
        pdf[i + half] = std::exp(-0.5f * x * x);
    }
    const int reach = kMeanRadius * steps + steps / 2 + sigmas * kStddevLen * steps;
    std::vector<float> smooth(2 * reach + 1);
    for (int i = -reach; i <= reach; ++i)
        smooth[i + reach] = SmoothScore(static_cast<float>(i) / steps, 0.0f, 1.0f, sayuri_go::kMaxBoard);
    for (int m = 0; m < kMeanLen; ++m) {
        const int mean_steps = (m - kMeanRadius) * steps - steps / 2;
        for (int s = 0; s < kStddevLen; ++s) {
            float w_sum = 0.0f, wv_sum = 0.0f;
            for (int i = -half; i <= half; ++i) {
                const float w = pdf[i + half];
                w_sum += w;
                wv_sum += w * smooth[mean_steps + s * i + reach];
            }
            table_[static_cast<size_t>(m) * kStddevLen + s] = wv_sum / w_sum;
        }
    }
}

const ScoreUtility& ScoreUtility::Get() {
    static const ScoreUtility u;
    return u;
}

float ScoreUtility::Expected(float mean, float stddev, float center, float scale, float board_size) const {
    const float k = static_cast<float>(sayuri_go::kMaxBoard) / (scale * board_size);
    const float mean_scaled = (mean - center) * k;
    const float stddev_scaled = stddev * k;
    const float mean_round = std::round(mean_scaled);
    const float stddev_floor = std::floor(stddev_scaled);
    int m0 = static_cast<int>(mean_round) + kMeanRadius;
    int s0 = static_cast<int>(stddev_floor);
    int m1 = m0 + 1, s1 = s0 + 1;
    if (m0 < 0) m0 = m1 = 0;
    if (m1 >= kMeanLen) m0 = m1 = kMeanLen - 1;
    if (s1 >= kStddevLen) s0 = s1 = kStddevLen - 1;
    const float lm = mean_scaled - mean_round + 0.5f;
    const float ls = stddev_scaled - stddev_floor;
    const float a00 = table_[static_cast<size_t>(m0) * kStddevLen + s0], a01 = table_[static_cast<size_t>(m0) * kStddevLen + s1];
    const float a10 = table_[static_cast<size_t>(m1) * kStddevLen + s0], a11 = table_[static_cast<size_t>(m1) * kStddevLen + s1];
    const float b0 = a00 + ls * (a01 - a00);
    const float b1 = a10 + ls * (a11 - a10);
    return b0 + lm * (b1 - b0);
}

// ---------------------------------------------------------------------------------------------
namespace {
double RationalApprox(double t) { // Abramowitz & Stegun 26.2.23
    const double c[3] = {2.515517, 0.802853, 0.010328};
    const double d[3] = {1.432788, 0.189269, 0.001308};
    return t - ((c[2] * t + c[1]) * t + c[0]) / (((d[2] * t + d[1]) * t + d[0]) * t + 1.0);
}
double NormalCdfInverse(double p) {
    if (p <= 0.0 || p >= 1.0) throw std::invalid_argument("NormalCdfInverse: p must lie in (0, 1)");
    if (p < 0.5) return -RationalApprox(std::sqrt(-2.0 * std::log(p)));
    return RationalApprox(std::sqrt(-2.0 * std::log(1 - p)));
}
double NormalToT(double z, double dof) { // KataGo fancymath approximation
    double n = dof + 2;
    if (dof > 8) {
        n -= 1;
        return std::sqrt(n * std::exp(z * z * (n - 1.5) / ((n - 1) * (n - 1))) - n);
    }
    return std::sqrt(n * std::exp(z * z * (n - 0.853999327911) / ((n - 1.044042304114) * (n - 0.954115472059))) - n);
}
} // namespace

TQuantiles::TQuantiles(float complement_probability) {
    const double z = NormalCdfInverse(1.0 - complement_probability);
    for (int i = 0; i < 1000; ++i) z_[static_cast<size_t>(i)] = static_cast<float>(NormalToT(z, i));
}

float TQuantiles::At(int v) const {
    if (v < 1) return z_[0];
    if (v < 1000) return z_[static_cast<size_t>(v - 1)];
    return z_[999];
}

} // namespace sayuri_engine
