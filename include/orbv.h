/* orbv.h — C ABI of the MI355X-native bag-of-words transform (part of liborbx.so).
 *
 * the step that follows descriptor extraction on every frame / keyframe.  The vocabulary tree
 * lives in HBM; descriptors produced by orbx_extract_batch_device are consumed where they are (device pointers).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   orbv_load_text          <- TemplatedVocabulary::loadFromTextFile(const std::string&)
 *                                  Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1345-1425   (called at src/main.cc:98)
 *   orbv_create             <- the same node table handed over as arrays (what loadFromTextFile parses per line:
 *                                  parent id, leaf flag, 32 descriptor bytes, weight)
 *   orbv_info               <- getBranchingFactor / getDepthLevels / getScoringType / getWeightingType / size()
 *                                  TemplatedVocabulary.h:185-223, :122
 *   orbv_descend[_device]   <- transform(const TDescriptor&, WordId&, WordValue&, NodeId*, int levelsup)
 *                                  TemplatedVocabulary.h:1218-1259, with FORB::distance  DBoW2/FORB.cpp:81-101
 *   orbv_transform[_batch_device]
 *                           <- transform(const std::vector<TDescriptor>&, BowVector&, FeatureVector&, int levelsup)
 *                                  TemplatedVocabulary.h:1127-1194; BowVector::addWeight / addIfNotExist / normalize
 *                                  DBoW2/BowVector.cpp:36-88; FeatureVector::addFeature  DBoW2/FeatureVector.cpp:32-47
 *                                  (called at src/Frame.cc:285 and src/KeyFrame.cc:63 with levelsup = 4)
 *   orbv_score              <- TemplatedVocabulary::score(const BowVector&, const BowVector&)  TemplatedVocabulary.h:1198-1203,
 *                                  DBoW2/ScoringObject.cpp (called at src/KeyFrameDatabase.cc:132,:248, src/LoopClosing.cc:127);
 *                                  a host function (a merge walk over two short sorted lists)
 *
 * Output forms.  A BowVector (std::map<WordId, WordValue>) is returned as two parallel arrays in ascending word
 * order; a FeatureVector (std::map<NodeId, std::vector<unsigned>>) as CSR: node ids ascending, fv_off[j]..fv_off[j+1]
 * delimits the feature indices of node j in fv_feat (ascending, the push_back order of the reference).  The CSR form
 * is exactly what orbm_match_top2_segments consumes for SearchByBoW-style candidate sets.
 * Values are bit-identical to the reference's doubles: sums run in feature order, norms in word order.
 *
 * Status codes are orbx.h's.  There is no CPU fallback: compute entry points return ORBX_ERR_DEVICE without a GPU.
 */
#ifndef ORBV_H
#define ORBV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DBoW2::WeightingType / ScoringType (Thirdparty/DBoW2/DBoW2/BowVector.h:36-53) */
#define ORBV_TF_IDF 0
#define ORBV_TF     1
#define ORBV_IDF    2
#define ORBV_BINARY 3
#define ORBV_L1_NORM       0
#define ORBV_L2_NORM       1
#define ORBV_CHI_SQUARE    2
#define ORBV_KL            3
#define ORBV_BHATTACHARYYA 4
#define ORBV_DOT_PRODUCT   5

#define ORBV_MAX_CHILDREN 32     /* children of one node (the text format admits k <= 20) */
#define ORBV_MAX_FEATURES 8192   /* features of one frame in orbv_transform* */

typedef struct orbv_vocabulary orbv_vocabulary;

/* Node table: node 0 is the root (its row is ignored); node i >= 1 has parent[i] < i, children keep table order,
 * word ids number the leaves in table order (as loadFromTextFile does).  is_leaf[i] must be set exactly for the
 * childless nodes.  Returns ORBX_ERR_ARG for an inconsistent table, ORBX_ERR_GEOMETRY for > ORBV_MAX_CHILDREN. */
int orbv_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf,
                const uint8_t* desc /* n_nodes x 32 */, const double* weight, int device, orbv_vocabulary** out);
/* The reference's text format: first line "k L scoring weighting", then one line per node
 * "parent isLeaf d0 .. d31 weight".  Same header validation as the reference (k<=20, 1<=L<=10, ...). */
int orbv_load_text(const char* path, int device, orbv_vocabulary** out);
void orbv_destroy(orbv_vocabulary* v);
int orbv_info(const orbv_vocabulary* v, int* k, int* L, int* scoring, int* weighting, int* n_words, int* n_nodes);

/* Per-descriptor tree descent: word id, word weight and the node id at level (L - levelsup) (0 = root when that
 * level is <= 0, or when the leaf lies above it — indeterminate in the reference).  Host / device pointers. */
int orbv_descend(const orbv_vocabulary* v, const uint8_t* desc, int n, int levelsup,
                 uint32_t* word, double* weight, uint32_t* node);
int orbv_descend_device(const orbv_vocabulary* v, const uint8_t* d_desc, int n, int levelsup,
                        uint32_t* d_word, double* d_weight, uint32_t* d_node, void* stream);

/* Frame::ComputeBoW for one frame, host pointers.  Capacities: bow_* and fv_node/fv_feat n entries, fv_off n+1. */
int orbv_transform(const orbv_vocabulary* v, const uint8_t* desc, int n, int levelsup,
                   uint32_t* bow_id, double* bow_val, int* n_bow,
                   uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* n_fv);
/* Throughput form, the layout orbx_extract_batch_device leaves behind: frame f has d_n[f] descriptors at
 * d_desc + f*cap*32.  Outputs of frame f at offset f*cap (fv_off: f*(cap+1)); counts in d_n_bow[f], d_n_fv[f].
 * cap <= ORBV_MAX_FEATURES.  Asynchronous on `stream` (hipStream_t); the handle owns scratch, so one call at a
 * time per vocabulary handle (clone handles for concurrent streams). */
int orbv_transform_batch_device(orbv_vocabulary* v, const uint8_t* d_desc, const int32_t* d_n, int nframes, int cap,
                                int levelsup, uint32_t* d_bow_id, double* d_bow_val, int32_t* d_n_bow,
                                uint32_t* d_fv_node, int32_t* d_fv_off, uint32_t* d_fv_feat, int32_t* d_n_fv,
                                void* stream);

/* score of two BowVectors under the vocabulary's scoring type (host, re-entrant) */
double orbv_score(const orbv_vocabulary* v, const uint32_t* id1, const double* val1, int n1,
                  const uint32_t* id2, const double* val2, int n2);

#ifdef __cplusplus
}
#endif
#endif
