// =====================================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp's header; the same rules apply).
//
// CPU restatement of the bag-of-words transform ORB_SLAM runs on every frame's descriptors
// (SURVEY.md §8f N1):
//   /root/reference/src/Frame.cc:280-287  Frame::ComputeBoW → ORBVocabulary::transform(desc, BowVec, FeatVec, 4)
//   /root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h
//       :1345-1425  loadFromTextFile   (node table, word ids = leaves in file order)
//       :1127-1194  transform(features, BowVector&, FeatureVector&, levelsup)
//       :1218-1259  transform(feature, word_id, weight, nid, levelsup)   (tree descent)
//   /root/reference/Thirdparty/DBoW2/DBoW2/BowVector.cpp:36-88  addWeight / addIfNotExist / normalize
//   /root/reference/Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:32-47  addFeature
//   /root/reference/Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101  distance
//   /root/reference/Thirdparty/DBoW2/DBoW2/ScoringObject.cpp  the six score() functions
//
// PARITY PINNED: unlike the extractor's OpenCV primitives, every operation on this path is DBoW2's
// own code, which is vendored in the reference tree.  oracle/Makefile compiles those sources where
// they lie (against oracle/cvstub, where cv::Mat is only a 32-byte container) into
// oracle/_ref/libref_dbow2.so, and tests/test_ref_pin.py checks this restatement against it on
// synthetic vocabularies (all weightings / scorings, ragged trees, stopped words).
// =====================================================================================
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {
enum { W_TF_IDF = 0, W_TF = 1, W_IDF = 2, W_BINARY = 3 };                                      // BowVector.h:36-42
enum { S_L1 = 0, S_L2 = 1, S_CHI = 2, S_KL = 3, S_BHATTA = 4, S_DOT = 5 };                     // BowVector.h:45-53

struct Node {
    int parent = 0;
    std::vector<int> children;
    uint8_t desc[32] = {0};
    double weight = 0;
    int word_id = 0;
};
struct Voc {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<Node> nodes;      // nodes[0] = root
    int nwords = 0;
};

// FORB.cpp:81-101 (the bit-count of src/ORBmatcher.cc:1794-1810 again)
int forb_distance(const uint8_t* a, const uint8_t* b) {
    const int32_t* pa = (const int32_t*)a;
    const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// mustNormalize of the scoring classes (ScoringObject.h:74-89): all but the dot product normalise; L2 only for L2
bool must_normalize(int scoring, int* norm_l2) {
    *norm_l2 = scoring == S_L2;
    return scoring != S_DOT;
}

// TemplatedVocabulary.h:1218-1259
void descend(const Voc& v, const uint8_t* f, int levelsup, unsigned* word_id, double* weight, unsigned* nid) {
    const int nid_level = v.L - levelsup;
    if (nid_level <= 0) *nid = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const std::vector<int>& ch = v.nodes[final_id].children;
        final_id = ch[0];
        double best_d = forb_distance(f, v.nodes[final_id].desc);
        for (size_t i = 1; i < ch.size(); i++) {
            double d = forb_distance(f, v.nodes[ch[i]].desc);
            if (d < best_d) { best_d = d; final_id = ch[i]; }
        }
        if (current_level == nid_level) *nid = final_id;
    } while (!v.nodes[final_id].children.empty());
    *word_id = v.nodes[final_id].word_id;
    *weight = v.nodes[final_id].weight;
}

typedef std::map<unsigned, double> BowVector;
typedef std::map<unsigned, std::vector<unsigned> > FeatureVector;

// TemplatedVocabulary.h:1127-1194 with BowVector.cpp:36-88 and FeatureVector.cpp:32-47 (std::map semantics)
void transform(const Voc& voc, const uint8_t* desc, int n, int levelsup, BowVector& v, FeatureVector& fv) {
    v.clear();
    fv.clear();
    if (voc.nwords == 0) return;
    int l2 = 0;
    const bool must = must_normalize(voc.scoring, &l2);
    const bool tf = voc.weighting == W_TF || voc.weighting == W_TF_IDF;
    for (int i = 0; i < n; i++) {
        unsigned id, nid = 0;      // the reference leaves nid indeterminate when a leaf sits above nid_level
        double w;
        descend(voc, desc + (size_t)i * 32, levelsup, &id, &w, &nid);
        if (w > 0) {
            if (tf) {
                BowVector::iterator it = v.find(id);
                if (it != v.end()) it->second += w; else v[id] = w;
            } else {
                if (v.find(id) == v.end()) v[id] = w;
            }
            fv[nid].push_back((unsigned)i);
        }
    }
    if (tf && !v.empty() && !must) {
        const double nd = (double)v.size();
        for (BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (!l2) { for (BowVector::iterator it = v.begin(); it != v.end(); ++it) norm += fabs(it->second); }
        else { for (BowVector::iterator it = v.begin(); it != v.end(); ++it) norm += it->second * it->second; norm = sqrt(norm); }
        if (norm > 0.0) for (BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    }
}

// ScoringObject.cpp: the merge walks (lower_bound jumps only skip keys that cannot match)
double score(int scoring, const unsigned* id1, const double* v1, int n1, const unsigned* id2, const double* v2, int n2) {
    const double LOG_EPS = log(DBL_EPSILON);
    int i = 0, j = 0;
    double s = 0;
    while (i < n1 && j < n2) {
        const double vi = v1[i], wi = v2[j];
        if (id1[i] == id2[j]) {
            switch (scoring) {
                case S_L1: s += fabs(vi - wi) - fabs(vi) - fabs(wi); break;
                case S_L2: case S_DOT: s += vi * wi; break;
                case S_CHI: if (vi + wi != 0.0) s += vi * wi / (vi + wi); break;
                case S_KL: if (vi != 0 && wi != 0) s += vi * log(vi / wi); break;
                case S_BHATTA: s += sqrt(vi * wi); break;
            }
            i++; j++;
        } else if (id1[i] < id2[j]) {
            if (scoring == S_KL) s += vi * (log(vi) - LOG_EPS);
            i++;
        } else {
            j++;
        }
    }
    switch (scoring) {
        case S_L1: return -s / 2.0;
        case S_L2: return s >= 1 ? 1.0 : 1.0 - sqrt(1.0 - s);
        case S_CHI: return 2. * s;
        case S_KL:
            for (; i < n1; i++) if (v1[i] != 0) s += v1[i] * (log(v1[i]) - LOG_EPS);
            return s;
        default: return s;
    }
}
}  // namespace

extern "C" {
// TemplatedVocabulary.h:1345-1425.  A well-formed file has no blank line at the end (the reference's
// `while(!f.eof())` loop would otherwise create a node with an indeterminate parent); blank lines end the table here.
void* orc_voc_load_text(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    std::string all;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) all.append(buf, got);
    fclose(f);
    Voc* v = new Voc();
    const char* p = all.c_str();
    char* e;
    v->k = (int)strtol(p, &e, 10); p = e;
    v->L = (int)strtol(p, &e, 10); p = e;
    int n1 = (int)strtol(p, &e, 10); p = e;
    int n2 = (int)strtol(p, &e, 10); p = e;
    if (v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) { delete v; return 0; }
    v->scoring = n1; v->weighting = n2;
    v->nodes.resize(1);
    while (*p && *p != '\n') p++;
    if (*p) p++;
    while (*p) {
        const char* eol = strchr(p, '\n');
        if (!eol) eol = p + strlen(p);
        const char* q = p;
        while (q < eol && (*q == ' ' || *q == '\r' || *q == '\t')) q++;
        if (q == eol) break;
        Node nd;
        nd.parent = (int)strtol(p, &e, 10); p = e;
        int leaf = (int)strtol(p, &e, 10); p = e;
        for (int i = 0; i < 32; i++) { nd.desc[i] = (uint8_t)strtol(p, &e, 10); p = e; }
        nd.weight = strtod(p, &e); p = e;
        const int nid = (int)v->nodes.size();
        if (nd.parent < 0 || nd.parent >= nid) { delete v; return 0; }
        if (leaf > 0) nd.word_id = v->nwords++;
        v->nodes.push_back(nd);
        v->nodes[nd.parent].children.push_back(nid);
        p = *eol ? eol + 1 : eol;
    }
    return v;
}
void* orc_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf,
                     const uint8_t* desc, const double* weight) {
    Voc* v = new Voc();
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->nodes.resize(n_nodes);
    for (int i = 1; i < n_nodes; i++) {
        Node& nd = v->nodes[i];
        nd.parent = parent[i];
        memcpy(nd.desc, desc + (size_t)i * 32, 32);
        nd.weight = weight[i];
        if (is_leaf[i]) nd.word_id = v->nwords++;
        v->nodes[parent[i]].children.push_back(i);
    }
    return v;
}
void orc_voc_destroy(void* h) { delete (Voc*)h; }
void orc_voc_info(void* h, int* k, int* L, int* scoring, int* weighting, int* nwords, int* nnodes) {
    Voc* v = (Voc*)h;
    *k = v->k; *L = v->L; *scoring = v->scoring; *weighting = v->weighting; *nwords = v->nwords; *nnodes = (int)v->nodes.size();
}
int orc_forb_distance(const uint8_t* a, const uint8_t* b) { return forb_distance(a, b); }
void orc_voc_descend(void* h, const uint8_t* desc, int n, int levelsup, unsigned* word, double* weight, unsigned* node) {
    for (int i = 0; i < n; i++) {
        node[i] = 0;
        descend(*(Voc*)h, desc + (size_t)i * 32, levelsup, word + i, weight + i, node + i);
    }
}
void orc_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, unsigned* bow_id, double* bow_val, int* n_bow,
                       unsigned* fv_node, int* fv_off, unsigned* fv_feat, int* n_fv) {
    BowVector bv;
    FeatureVector fv;
    transform(*(Voc*)h, desc, n, levelsup, bv, fv);
    int i = 0;
    for (BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++i) { bow_id[i] = it->first; bow_val[i] = it->second; }
    *n_bow = i;
    int j = 0, o = 0;
    fv_off[0] = 0;
    for (FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++j) {
        fv_node[j] = it->first;
        for (size_t q = 0; q < it->second.size(); q++) fv_feat[o++] = it->second[q];
        fv_off[j + 1] = o;
    }
    *n_fv = j;
}
double orc_voc_score(void* h, const unsigned* id1, const double* v1, int n1, const unsigned* id2, const double* v2, int n2) {
    return score(((Voc*)h)->scoring, id1, v1, n1, id2, v2, n2);
}
}
