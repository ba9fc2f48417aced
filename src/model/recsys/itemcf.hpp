#ifndef CDAE_HOST_MODEL_RECSYS_ITEMCF_HPP_
#define CDAE_HOST_MODEL_RECSYS_ITEMCF_HPP_
#include <model/recsys/similarity_base.hpp>
namespace libcf {
class ItemCF : public SimilarityBase {   // reference: src/model/recsys/itemcf.hpp:10-19 (out of scope, see similarity_base.hpp)
 public:
  ItemCF(SimilarityType st = Jaccard, size_t topk = 50) : SimilarityBase(1, 0, st, topk) {}
};
}  // namespace libcf
#endif
