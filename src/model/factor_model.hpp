// FactorModel — generic factorisation machine of the reference (src/model/factor_model.hpp), included by
// apps/yelp but never instantiated there.  OUT OF SCOPE: declaration only.
#ifndef CDAE_HOST_MODEL_FACTOR_MODEL_HPP_
#define CDAE_HOST_MODEL_FACTOR_MODEL_HPP_
#include <model/model_base.hpp>
namespace libcf {
struct FactorModelConfig { double lambda = 0.; LossType lt = SQUARE; PenaltyType pt = L2; size_t num_dim = 10; };
class FactorModel : public ModelBase, public SGDBase {
 public:
  FactorModel() = default;
  explicit FactorModel(const FactorModelConfig&) {}
};
}  // namespace libcf
#endif
