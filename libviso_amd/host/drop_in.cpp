// drop_in.cpp — see drop_in.hpp.
#include "drop_in.hpp"

#include <cstring>
#include <string>

#include "kitti_shard.hpp"
#include "viso.hpp"

namespace viso { void set_host_error(const std::string& s); }

extern "C" int viso_host_drop_in_run(const float* kp, const float* desc, const int32_t* n, int nf, int cap, int dlen,
                                     const double F[9], const viso_param* prm, uint64_t ransac_seed, uint64_t first_frame,
                                     double* rec8, int32_t* matches, int32_t* match_n, int32_t* n_circle, double* call_us,
                                     double* loop_s) {
    using namespace viso;
    if (!kp || !desc || !n || nf < 0 || cap < 1 || dlen < 1 || !F || !prm) { set_host_error("viso_host_drop_in_run: bad argument"); return VISO_ERR_ARG; }
    try {
        // the frames as the front end hands them over (detector.detect + extractor.compute, src/viso.cpp:1226-1231), built
        // before the loop starts: the loop's clock sees the hot path, not this reshaping
        std::vector<StereoFeatures> frames((size_t)nf);
        for (int t = 0; t < nf; ++t)
            for (int side = 0; side < 2; ++side) {
                const int k = n[(size_t)t * 2 + side];
                if (k < 0 || k > cap) { set_host_error("viso_host_drop_in_run: n out of range"); return VISO_ERR_ARG; }
                KeyPoints& kk = side ? frames[(size_t)t].kp2 : frames[(size_t)t].kp1;
                Descriptors& dd = side ? frames[(size_t)t].d2 : frames[(size_t)t].d1;
                const float* ks = kp + ((size_t)t * 2 + side) * cap * 2;
                kk.resize((size_t)k);
                for (int i = 0; i < k; ++i) { kk[(size_t)i].pt.x = ks[2 * i]; kk[(size_t)i].pt.y = ks[2 * i + 1]; }
                dd.rows = k; dd.cols = dlen;
                dd.data.assign(desc + ((size_t)t * 2 + side) * cap * dlen, desc + ((size_t)t * 2 + side) * cap * dlen + (size_t)k * dlen);
            }
        Matd Fm(3, 3);
        for (int i = 0; i < 9; ++i) Fm.data[(size_t)i] = F[i];
        param p;
        p.base = prm->base; p.ransac_iter = prm->ransac_iter; p.inlier_threshold = prm->inlier_threshold;
        p.thresh = prm->thresh; p.save_debug = false;
        p.calib.f = prm->f; p.calib.cu = prm->cu; p.calib.cv = prm->cv;
        p.ransac_seed = ransac_seed;
        size_t next = 0;
        StereoFeatureGenerator gen = [&]() -> std::optional<StereoFeatures> {
            if (next >= frames.size()) return std::nullopt;
            return std::move(frames[next++]);
        };
        PerCallStats st;
        PerCallTrace tr;
        const bool want_trace = matches || match_n || n_circle;
        OdometryResult out = sequence_odometry_per_call(Fm, p, gen, first_frame, &st, want_trace ? &tr : nullptr);
        const int done = st.frames;
        if (rec8) {
            std::memset(rec8, 0, sizeof(double) * 8 * (size_t)nf);
            for (int t = 0; t < done; ++t) {
                for (int j = 0; j < 6; ++j) rec8[(size_t)t * 8 + j] = out.tr[(size_t)t][(size_t)j];
                rec8[(size_t)t * 8 + 6] = out.ok[(size_t)t];
                rec8[(size_t)t * 8 + 7] = out.n_inliers[(size_t)t];
            }
        }
        if (want_trace)
            for (int which = 0; which < 3; ++which)
                for (int t = 0; t < done; ++t) {
                    const Matches& m = which == 0 ? tr.match_lr[(size_t)t] : which == 1 ? tr.match11[(size_t)t] : tr.match22[(size_t)t];
                    if (match_n) match_n[(size_t)which * nf + t] = (int32_t)m.size();
                    if (matches) {
                        int32_t* dst = matches + ((size_t)which * nf + t) * cap * 3;
                        for (size_t i = 0; i < m.size() && i < (size_t)cap; ++i) { dst[3 * i] = m[i][0]; dst[3 * i + 1] = m[i][1]; dst[3 * i + 2] = m[i][2]; }
                    }
                }
        if (n_circle) for (int t = 0; t < done; ++t) n_circle[t] = tr.n_circle[(size_t)t];
        if (call_us) for (int f = 0; f < VISO_PLAIN_N; ++f) { call_us[2 * f] = (double)st.calls[f]; call_us[2 * f + 1] = st.us[f]; }
        if (loop_s) { loop_s[0] = st.wall_s; loop_s[1] = st.carry_s; loop_s[2] = 0; }
        return done;
    } catch (const std::invalid_argument& e) {
        set_host_error(std::string("viso_host_drop_in_run: ") + e.what());
        return VISO_ERR_ARG;
    } catch (const std::exception& e) {
        set_host_error(std::string("viso_host_drop_in_run: ") + e.what());
        return VISO_ERR_HIP;
    }
}
