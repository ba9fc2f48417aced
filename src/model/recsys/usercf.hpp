#ifndef CDAE_HOST_MODEL_RECSYS_USERCF_HPP_
#define CDAE_HOST_MODEL_RECSYS_USERCF_HPP_
#include <model/recsys/similarity_base.hpp>
namespace libcf {
class UserCF : public SimilarityBase {   // reference: src/model/recsys/usercf.hpp (out of scope, see similarity_base.hpp)
 public:
  UserCF(SimilarityType st = Jaccard, size_t topk = 50) : SimilarityBase(0, 1, st, topk) {}
};
}  // namespace libcf
#endif
