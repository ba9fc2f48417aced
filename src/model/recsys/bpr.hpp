// BPR — OUT OF SCOPE of the MI355X hot path (see imf.hpp).  Reference: src/model/recsys/bpr.hpp:12-50.
#ifndef CDAE_HOST_MODEL_RECSYS_BPR_HPP_
#define CDAE_HOST_MODEL_RECSYS_BPR_HPP_

#include <model/recsys/imf.hpp>

namespace libcf {

struct BPRConfig {
  double learn_rate = 0.1;
  double beta = 1.;
  double lambda = 0.01;
  LossType lt = LOG;
  PenaltyType pt = L2;
  size_t num_dim = 10;
  size_t num_neg = 5;
  bool using_bias_term = true;
  bool using_adagrad = true;
};

class BPR : public IMF {
 public:
  BPR() = default;
  explicit BPR(const BPRConfig&) {}
  void reset(const Data&) {
    LOG(FATAL) << "--method=BPR is not provided by this build: only the CDAE training hot path "
                  "(--method=CDAE) and the Popularity baseline are (SURVEY.md §2.1)";
  }
};

}  // namespace libcf
#endif
