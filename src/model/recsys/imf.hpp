// IMF (implicit matrix factorisation) — OUT OF SCOPE of the MI355X hot path.
// The config struct and class keep the reference's names and fields (src/model/recsys/imf.hpp:12-54) so that
// apps/yelp compiles unchanged; selecting --method=MF aborts with a clear message.
#ifndef CDAE_HOST_MODEL_RECSYS_IMF_HPP_
#define CDAE_HOST_MODEL_RECSYS_IMF_HPP_

#include <model/recsys/recsys_model_base.hpp>

namespace libcf {

struct IMFConfig {
  double learn_rate = 0.1;
  double beta = 1.;
  double lambda = 0.01;
  LossType lt = SQUARE;
  PenaltyType pt = L2;
  size_t num_dim = 10;
  size_t num_neg = 5;
  bool using_bias_term = true;
  bool using_adagrad = true;
};

class IMF : public RecsysModelBase {
 public:
  IMF() = default;
  explicit IMF(const IMFConfig& cfg) : cfg_(cfg) {}
  void reset(const Data&) {
    LOG(FATAL) << "--method=MF (IMF) is not provided by this build: only the CDAE training hot path "
                  "(--method=CDAE) and the Popularity baseline are (SURVEY.md §2.1)";
  }
  void train_one_iteration(const Data&) {}
 protected:
  IMFConfig cfg_;
};

}  // namespace libcf
#endif
