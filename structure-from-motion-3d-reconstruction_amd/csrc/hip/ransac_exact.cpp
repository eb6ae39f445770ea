// ransac_exact.cpp — the reference's eight_point_E (cpp/src/templering_sfm.cpp T:609-627, via AtA_from_A T:503-517,
// jacobi_eig_sym linalg.hpp:133-201 with the platform's atan2/cos/sin, enforce_rank2 T:595-607) for a list of RANSAC
// iterations, on a host worker team.  Part of libsfmx.so (compiled by g++ with -ffp-contract=off, like the rest of the
// host math): sfmx_ransac_score_ex patches these rows over the device's hypotheses where the device's libm-free Jacobi
// cannot be trusted to land on the same vector as the reference -- octets with a repeated sample index (null space of
// dimension >= 2) and ill-conditioned ones.
#include <cstdint>
#include <cstring>

#include "../host/host_math.hpp"
#include "../host/thread_pool.hpp"

extern "C" void sfmx_exact_eight_point_batch(const double* xi, const double* xj, const std::int32_t* idx8, const std::int32_t* iters, int m,
                                             double* E_out) {
  auto one = [&](int k) {
    const std::int32_t* o = idx8 + (std::size_t)8 * (std::size_t)iters[k];
    const int oct[8] = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
    const sfmx_host::Mat3 E = sfmx_host::eight_point_E(xi, xj, oct);
    std::memcpy(E_out + (std::size_t)9 * (std::size_t)k, E.a, 72);
  };
  sfmx_host::ThreadPool::instance().parallel_for(m, one, 4);
}
