// Bag-of-words transform on gfx950: ORBVocabulary::transform(descriptors, BowVector, FeatureVector,
// levelsup) of reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194 / :1218-1259, as called per frame at
// src/Frame.cc:285.  include/orbv.h is the boundary.
//
// Data layout in HBM.  The node table is renumbered breadth-first so that the children of a node are contiguous:
//   first_child[i] .. first_child[i+1]  = the (internal) ids of node i's children, in the reference's child order
//   desc[i]      32 B descriptor of internal node i (so one descent level reads ONE contiguous k*32-byte block)
//   orig_id[i]   the reference's NodeId; word[i] / weight[i] the reference's WordId / weight (leaves)
// A k=10, L=6 vocabulary is 1.11 M nodes = 35.6 MB of descriptors: it sits in the 256 MiB Infinity Cache, its first
// four levels (11 k nodes, 355 KB) in every XCD's L2.
//
// Kernels (integer/bitwise + a little f64; no MFMA):
//   k_descend<G>   G = 16 or 32 lanes per descriptor, one lane per CHILD: each lane loads its child's 32 bytes (the
//                  group's loads are one coalesced k*32 B block), xor+popcount against the query held in 8 VGPRs, and
//                  the group takes the minimum of (distance << 8 | child) — the first child attaining the smallest
//                  distance, i.e. the reference's strict `d < best_d` scan.  L dependent steps per descriptor; the
//                  launch carries every descriptor of every frame so the latency is hidden by occupancy.
//   k_assemble     one workgroup per frame: bitonic sort of (word << 32 | feature) in LDS → BowVector (sum of the
//                  word's weight in feature order, exactly the `+=` chain of BowVector::addWeight), the normalisation
//                  as ONE sequential f64 sum in ascending word order (the std::map iteration order — parallel
//                  summation would change the rounding), then the same sort on (node << 32 | feature) → FeatureVector
//                  as CSR.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "orbv.h"
#include "orbx.h"

namespace orbv {

struct VocDev {                     // by value in the kernarg segment
    const int32_t* first_child;     // n_nodes + 1
    const uint4* desc;              // n_nodes x 2
    const uint32_t* orig_id;
    const uint32_t* word;
    const double* weight;
    int L;
};

constexpr int DESC_BLOCK = 256;
constexpr int ASM_BLOCK = 256;

template <int G>
__global__ __launch_bounds__(DESC_BLOCK) void k_descend(VocDev v, const uint8_t* __restrict__ desc, const int32_t* __restrict__ d_n,
                                                       int n_or_cap, int levelsup, uint32_t* __restrict__ word,
                                                       double* __restrict__ weight, uint32_t* __restrict__ node) {
    const int frame = blockIdx.y;
    const int n = d_n ? min(d_n[frame], n_or_cap) : n_or_cap;
    const int c = threadIdx.x % G;
    const int i = blockIdx.x * (DESC_BLOCK / G) + threadIdx.x / G;
    if (i >= n) return;
    const size_t slot = (size_t)frame * n_or_cap + i;
    const uint4* qp = (const uint4*)(desc + slot * 32);
    const uint4 q0 = qp[0], q1 = qp[1];
    const int nid_level = v.L - levelsup;
    int cur = 0, level = 0, nid = 0;
    for (;;) {
        const int fc = v.first_child[cur];
        const int nc = v.first_child[cur + 1] - fc;
        if (nc == 0) break;
        ++level;
        uint32_t key = 0xFFFFFFFFu;
        if (c < nc) {
            const uint4 a = v.desc[(size_t)(fc + c) * 2], b = v.desc[(size_t)(fc + c) * 2 + 1];
            const uint32_t d = __popc(a.x ^ q0.x) + __popc(a.y ^ q0.y) + __popc(a.z ^ q0.z) + __popc(a.w ^ q0.w) +
                               __popc(b.x ^ q1.x) + __popc(b.y ^ q1.y) + __popc(b.z ^ q1.z) + __popc(b.w ^ q1.w);
            key = (d << 8) | (uint32_t)c;
        }
#pragma unroll
        for (int m = G / 2; m > 0; m >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, m, G));
        cur = fc + (int)(key & 255u);
        if (level == nid_level) nid = cur;
    }
    if (c == 0) {
        word[slot] = v.word[cur];
        weight[slot] = v.weight[cur];
        node[slot] = v.orig_id[nid];
    }
}

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* key, int P) {
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += ASM_BLOCK) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key[i], b = key[ixj];
                    if ((a > b) == ((i & k) == 0)) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// After the sort: key[0..m) valid (ascending), the rest ~0.  Every thread owns a contiguous chunk; returns through
// LDS the output slot of each segment head (slot_of_chunk[t] = heads before chunk t) and the head count.
struct AsmShared {
    int chunk_heads[ASM_BLOCK];
    int total_heads;
    int m;
    double norm;
};

__device__ __forceinline__ int count_valid_and_heads(const unsigned long long* key, int P, AsmShared* sh, int& lo, int& hi) {
    const int C = (P + ASM_BLOCK - 1) / ASM_BLOCK;
    lo = threadIdx.x * C;
    hi = min(lo + C, P);
    int heads = 0, valid = 0;
    for (int i = lo; i < hi; i++) {
        const unsigned long long k = key[i];
        if (k == ~0ull) break;
        valid++;
        if (i == 0 || (uint32_t)(key[i - 1] >> 32) != (uint32_t)(k >> 32)) heads++;
    }
    sh->chunk_heads[threadIdx.x] = heads;
    if (threadIdx.x == 0) sh->m = 0;
    __syncthreads();
    if (valid) atomicAdd(&sh->m, valid);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < ASM_BLOCK; t++) { const int h = sh->chunk_heads[t]; sh->chunk_heads[t] = run; run += h; }
        sh->total_heads = run;
    }
    __syncthreads();
    return sh->chunk_heads[threadIdx.x];
}

// weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; norm_mode: 0 none (dot product), 1 L1, 2 L2
__global__ __launch_bounds__(ASM_BLOCK) void k_assemble(const int32_t* __restrict__ d_n, int n_or_cap, int P, int weighting, int norm_mode,
                                                       const uint32_t* __restrict__ word, const double* __restrict__ weight,
                                                       const uint32_t* __restrict__ node, uint32_t* __restrict__ bow_id,
                                                       double* __restrict__ bow_val, int32_t* __restrict__ n_bow,
                                                       uint32_t* __restrict__ fv_node, int32_t* __restrict__ fv_off,
                                                       uint32_t* __restrict__ fv_feat, int32_t* __restrict__ n_fv) {
    extern __shared__ unsigned long long lds64[];
    unsigned long long* key = lds64;                 // P
    double* val = (double*)(lds64 + P);              // P
    __shared__ AsmShared sh;
    const int frame = blockIdx.x;
    const int n = d_n ? min(d_n[frame], n_or_cap) : n_or_cap;
    const size_t base = (size_t)frame * n_or_cap;
    word += base; weight += base; node += base;
    bow_id += base; bow_val += base; fv_node += base; fv_feat += base;
    fv_off += (size_t)frame * (n_or_cap + 1);
    const bool tf = weighting == ORBV_TF_IDF || weighting == ORBV_TF;

    // ---- BowVector
    for (int i = threadIdx.x; i < P; i += ASM_BLOCK)
        key[i] = (i < n && weight[i] > 0) ? (((unsigned long long)word[i] << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    bitonic_sort_u64(key, P);
    int lo, hi;
    int slot = count_valid_and_heads(key, P, &sh, lo, hi);
    const int m = sh.m, nb = sh.total_heads;
    for (int i = lo; i < hi && i < m; i++) {
        const uint32_t w = (uint32_t)(key[i] >> 32);
        if (i == 0 || (uint32_t)(key[i - 1] >> 32) != w) {
            const double wt = weight[(uint32_t)key[i]];
            double acc = wt;                                       // BowVector.cpp:36-48 (insert, then `+=` per further feature)
            if (tf) for (int j = i + 1; j < m && (uint32_t)(key[j] >> 32) == w; j++) acc += wt;
            bow_id[slot] = w;
            val[slot] = acc;
            slot++;
        }
    }
    __syncthreads();
    if (tf && norm_mode == 0 && nb > 0) {                          // TemplatedVocabulary.h:1163-1169
        const double nd = (double)nb;
        for (int s = threadIdx.x; s < nb; s += ASM_BLOCK) val[s] /= nd;
        __syncthreads();
    }
    if (norm_mode != 0) {                                          // BowVector.cpp:63-88, map (ascending word) order
        if (threadIdx.x == 0) {
            double norm = 0.0;
            if (norm_mode == 1) { for (int s = 0; s < nb; s++) norm += fabs(val[s]); }
            else { for (int s = 0; s < nb; s++) norm += val[s] * val[s]; norm = sqrt(norm); }
            sh.norm = norm;
        }
        __syncthreads();
        const double norm = sh.norm;
        if (norm > 0.0) for (int s = threadIdx.x; s < nb; s += ASM_BLOCK) val[s] /= norm;
        __syncthreads();
    }
    for (int s = threadIdx.x; s < nb; s += ASM_BLOCK) bow_val[s] = val[s];
    if (threadIdx.x == 0) n_bow[frame] = nb;
    __syncthreads();

    // ---- FeatureVector (FeatureVector.cpp:32-47): CSR over ascending node id, features ascending within a node
    for (int i = threadIdx.x; i < P; i += ASM_BLOCK)
        key[i] = (i < n && weight[i] > 0) ? (((unsigned long long)node[i] << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    bitonic_sort_u64(key, P);
    slot = count_valid_and_heads(key, P, &sh, lo, hi);
    const int mf = sh.m, nf = sh.total_heads;
    for (int i = lo; i < hi && i < mf; i++) {
        const uint32_t nd = (uint32_t)(key[i] >> 32);
        fv_feat[i] = (uint32_t)key[i];
        if (i == 0 || (uint32_t)(key[i - 1] >> 32) != nd) {
            fv_node[slot] = nd;
            fv_off[slot] = i;
            slot++;
        }
    }
    if (threadIdx.x == 0) { fv_off[nf] = mf; n_fv[frame] = nf; }
}

}  // namespace orbv

// ------------------------------------------------------------------------------------------------ host side
struct orbv_vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0, n_words = 0, n_nodes = 0, max_children = 0, device = 0;
    orbv::VocDev dev{};
    void* d_block = nullptr;              // one allocation holding the five tables
    // scratch of the per-feature descent results (transform paths)
    uint32_t* s_word = nullptr;
    double* s_weight = nullptr;
    uint32_t* s_node = nullptr;
    size_t s_cap = 0;
};

namespace {
struct HostNode {
    int parent;
    uint8_t leaf;
    uint8_t desc[32];
    double weight;
};

int build(int k, int L, int scoring, int weighting, const std::vector<HostNode>& nodes, int device, orbv_vocabulary** out) {
    const int n = (int)nodes.size();
    if (!out || n < 1 || L < 1 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return ORBX_ERR_ARG;
    std::vector<int> nchild(n, 0);
    for (int i = 1; i < n; i++) {
        if (nodes[i].parent < 0 || nodes[i].parent >= i) return ORBX_ERR_ARG;
        nchild[nodes[i].parent]++;
    }
    int max_children = 0;
    for (int i = 0; i < n; i++) {
        if (i > 0 && (nodes[i].leaf != 0) != (nchild[i] == 0)) return ORBX_ERR_ARG;
        max_children = std::max(max_children, nchild[i]);
    }
    if (max_children > ORBV_MAX_CHILDREN) return ORBX_ERR_GEOMETRY;
    // child lists in table order
    std::vector<int> coff(n + 1, 0), clist(std::max(n - 1, 0)), fill(n, 0);
    for (int i = 0; i < n; i++) coff[i + 1] = coff[i] + nchild[i];
    for (int i = 1; i < n; i++) clist[coff[nodes[i].parent] + fill[nodes[i].parent]++] = i;
    // word ids: leaves in table order (TemplatedVocabulary.h:1408-1414)
    std::vector<uint32_t> word_of(n, 0);
    int n_words = 0;
    for (int i = 1; i < n; i++) if (nodes[i].leaf) word_of[i] = (uint32_t)n_words++;
    // breadth-first renumbering: children of a node become contiguous
    std::vector<int> order;        // internal -> table id
    order.reserve(n);
    order.push_back(0);
    std::vector<int32_t> first_child(n + 1, 0);
    size_t head = 0;
    while (head < order.size()) {
        const int t = order[head];
        first_child[head] = (int32_t)order.size();
        for (int j = coff[t]; j < coff[t + 1]; j++) order.push_back(clist[j]);
        head++;
    }
    if ((int)order.size() != n) return ORBX_ERR_ARG;     // unreachable nodes cannot happen with parent < i, kept as a guard
    first_child[n] = n;
    std::vector<uint8_t> desc((size_t)n * 32);
    std::vector<uint32_t> orig(n), word(n);
    std::vector<double> weight(n);
    for (int i = 0; i < n; i++) {
        const int t = order[i];
        memcpy(&desc[(size_t)i * 32], nodes[t].desc, 32);
        orig[i] = (uint32_t)t;
        word[i] = word_of[t];
        weight[i] = nodes[t].weight;
    }
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_desc = 0, o_fc = al((size_t)n * 32), o_orig = o_fc + al((size_t)(n + 1) * 4), o_word = o_orig + al((size_t)n * 4),
                 o_wt = o_word + al((size_t)n * 4), total = o_wt + al((size_t)n * 8);
    orbv_vocabulary* v = new orbv_vocabulary();
    if (hipMalloc(&v->d_block, total) != hipSuccess) { delete v; return ORBX_ERR_DEVICE; }
    char* b = (char*)v->d_block;
    if (hipMemcpy(b + o_desc, desc.data(), (size_t)n * 32, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + o_fc, first_child.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + o_orig, orig.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + o_word, word.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + o_wt, weight.data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(v->d_block);
        delete v;
        return ORBX_ERR_DEVICE;
    }
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_words = n_words; v->n_nodes = n;
    v->max_children = max_children; v->device = device;
    v->dev.desc = (const uint4*)(b + o_desc);
    v->dev.first_child = (const int32_t*)(b + o_fc);
    v->dev.orig_id = (const uint32_t*)(b + o_orig);
    v->dev.word = (const uint32_t*)(b + o_word);
    v->dev.weight = (const double*)(b + o_wt);
    v->dev.L = L;
    *out = v;
    return ORBX_OK;
}

int launch_descend(const orbv_vocabulary* v, const uint8_t* d_desc, const int32_t* d_n, int n_or_cap, int nframes, int levelsup,
                   uint32_t* d_word, double* d_weight, uint32_t* d_node, hipStream_t st) {
    if (v->n_words == 0 || v->n_nodes < 2) return ORBX_ERR_ARG;     // the reference returns early on an empty vocabulary
    if (v->max_children <= 16) {
        dim3 grid((n_or_cap + 15) / 16, nframes);
        hipLaunchKernelGGL(orbv::k_descend<16>, grid, dim3(orbv::DESC_BLOCK), 0, st, v->dev, d_desc, d_n, n_or_cap, levelsup, d_word, d_weight, d_node);
    } else {
        dim3 grid((n_or_cap + 7) / 8, nframes);
        hipLaunchKernelGGL(orbv::k_descend<32>, grid, dim3(orbv::DESC_BLOCK), 0, st, v->dev, d_desc, d_n, n_or_cap, levelsup, d_word, d_weight, d_node);
    }
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int ensure_scratch(orbv_vocabulary* v, size_t slots) {
    if (slots <= v->s_cap) return ORBX_OK;
    if (v->s_word) { (void)hipFree(v->s_word); (void)hipFree(v->s_weight); (void)hipFree(v->s_node); v->s_word = nullptr; v->s_cap = 0; }
    if (hipMalloc(&v->s_word, slots * 4) != hipSuccess || hipMalloc(&v->s_weight, slots * 8) != hipSuccess ||
        hipMalloc(&v->s_node, slots * 4) != hipSuccess)
        return ORBX_ERR_DEVICE;
    v->s_cap = slots;
    return ORBX_OK;
}
}  // namespace

extern "C" {

int orbv_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf,
                const uint8_t* desc, const double* weight, int device, orbv_vocabulary** out) {
    if (!parent || !is_leaf || !desc || !weight || n_nodes < 1) return ORBX_ERR_ARG;
    std::vector<HostNode> nodes(n_nodes);
    nodes[0] = HostNode{0, 0, {0}, 0.0};
    for (int i = 1; i < n_nodes; i++) {
        nodes[i].parent = parent[i];
        nodes[i].leaf = is_leaf[i] ? 1 : 0;
        memcpy(nodes[i].desc, desc + (size_t)i * 32, 32);
        nodes[i].weight = weight[i];
    }
    return build(k, L, scoring, weighting, nodes, device, out);
}

int orbv_load_text(const char* path, int device, orbv_vocabulary** out) {
    if (!path || !out) return ORBX_ERR_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) return ORBX_ERR_ARG;
    std::string text;
    {
        std::vector<char> buf(1 << 20);
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) text.append(buf.data(), got);
    }
    fclose(f);
    const char* p = text.c_str();
    char* e = nullptr;
    const long k = strtol(p, &e, 10); p = e;
    const long L = strtol(p, &e, 10); p = e;
    const long sc = strtol(p, &e, 10); p = e;
    const long wt = strtol(p, &e, 10); p = e;
    // TemplatedVocabulary.h:1366-1370
    if (k < 0 || k > 20 || L < 1 || L > 10 || sc < 0 || sc > 5 || wt < 0 || wt > 3) return ORBX_ERR_ARG;
    std::vector<HostNode> nodes(1);
    nodes[0] = HostNode{0, 0, {0}, 0.0};
    while (*p && *p != '\n') p++;
    while (*p) {
        while (*p == '\n' || *p == '\r' || *p == ' ' || *p == '\t') p++;
        if (!*p) break;
        HostNode nd;
        nd.parent = (int)strtol(p, &e, 10);
        if (e == p) return ORBX_ERR_ARG;
        p = e;
        nd.leaf = strtol(p, &e, 10) > 0 ? 1 : 0; p = e;
        for (int i = 0; i < 32; i++) { nd.desc[i] = (uint8_t)strtol(p, &e, 10); p = e; }
        const char* before = p;
        nd.weight = strtod(p, &e);
        if (e == before) return ORBX_ERR_ARG;
        p = e;
        nodes.push_back(nd);
        while (*p && *p != '\n') p++;
    }
    return build((int)k, (int)L, (int)sc, (int)wt, nodes, device, out);
}

void orbv_destroy(orbv_vocabulary* v) {
    if (!v) return;
    if (v->d_block) (void)hipFree(v->d_block);
    if (v->s_word) { (void)hipFree(v->s_word); (void)hipFree(v->s_weight); (void)hipFree(v->s_node); }
    delete v;
}

int orbv_info(const orbv_vocabulary* v, int* k, int* L, int* scoring, int* weighting, int* n_words, int* n_nodes) {
    if (!v) return ORBX_ERR_ARG;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (n_words) *n_words = v->n_words;
    if (n_nodes) *n_nodes = v->n_nodes;
    return ORBX_OK;
}

int orbv_descend_device(const orbv_vocabulary* v, const uint8_t* d_desc, int n, int levelsup, uint32_t* d_word, double* d_weight,
                        uint32_t* d_node, void* stream) {
    if (!v || n < 0 || (n > 0 && (!d_desc || !d_word || !d_weight || !d_node))) return ORBX_ERR_ARG;
    if (n == 0) return ORBX_OK;
    if (hipSetDevice(v->device) != hipSuccess) return ORBX_ERR_DEVICE;
    return launch_descend(v, d_desc, nullptr, n, 1, levelsup, d_word, d_weight, d_node, (hipStream_t)stream);
}

int orbv_descend(const orbv_vocabulary* v, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node) {
    if (!v || n < 0 || (n > 0 && (!desc || !word || !weight || !node))) return ORBX_ERR_ARG;
    if (n == 0) return ORBX_OK;
    if (hipSetDevice(v->device) != hipSuccess) return ORBX_ERR_DEVICE;
    uint8_t* d = nullptr;
    int rc = ORBX_ERR_DEVICE;
    const size_t o_w = (size_t)n * 32, o_wt = o_w + (((size_t)n * 4 + 7) & ~(size_t)7), o_n = o_wt + (size_t)n * 8;
    if (hipMalloc(&d, o_n + (size_t)n * 4) == hipSuccess && hipMemcpy(d, desc, (size_t)n * 32, hipMemcpyHostToDevice) == hipSuccess) {
        rc = launch_descend(v, d, nullptr, n, 1, levelsup, (uint32_t*)(d + o_w), (double*)(d + o_wt), (uint32_t*)(d + o_n), nullptr);
        if (rc == ORBX_OK && (hipMemcpy(word, d + o_w, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(weight, d + o_wt, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(node, d + o_n, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = ORBX_ERR_DEVICE;
    }
    if (d) (void)hipFree(d);
    return rc;
}

int orbv_transform_batch_device(orbv_vocabulary* v, const uint8_t* d_desc, const int32_t* d_n, int nframes, int cap, int levelsup,
                                uint32_t* d_bow_id, double* d_bow_val, int32_t* d_n_bow, uint32_t* d_fv_node, int32_t* d_fv_off,
                                uint32_t* d_fv_feat, int32_t* d_n_fv, void* stream) {
    if (!v || nframes < 0 || cap < 1 || cap > ORBV_MAX_FEATURES) return ORBX_ERR_ARG;
    if (nframes == 0) return ORBX_OK;
    if (!d_desc || !d_bow_id || !d_bow_val || !d_n_bow || !d_fv_node || !d_fv_off || !d_fv_feat || !d_n_fv) return ORBX_ERR_ARG;
    if (hipSetDevice(v->device) != hipSuccess) return ORBX_ERR_DEVICE;
    int rc = ensure_scratch(v, (size_t)nframes * cap);
    if (rc != ORBX_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    rc = launch_descend(v, d_desc, d_n, cap, nframes, levelsup, v->s_word, v->s_weight, v->s_node, st);
    if (rc != ORBX_OK) return rc;
    int P = 1;
    while (P < cap) P <<= 1;
    const int norm_mode = v->scoring == ORBV_DOT_PRODUCT ? 0 : (v->scoring == ORBV_L2_NORM ? 2 : 1);   // ScoringObject.h:74-89
    const size_t lds = (size_t)P * 16;
    // the dynamic-LDS limit is an attribute of the CURRENT device: set per launch, not once per process
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)orbv::k_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, ORBV_MAX_FEATURES * 16) != hipSuccess)
        return ORBX_ERR_DEVICE;
    hipLaunchKernelGGL(orbv::k_assemble, dim3(nframes), dim3(orbv::ASM_BLOCK), lds, st, d_n, cap, P, v->weighting, norm_mode,
                       v->s_word, v->s_weight, v->s_node, d_bow_id, d_bow_val, d_n_bow, d_fv_node, d_fv_off, d_fv_feat, d_n_fv);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbv_transform(const orbv_vocabulary* v, const uint8_t* desc, int n, int levelsup, uint32_t* bow_id, double* bow_val, int* n_bow,
                   uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int* n_fv) {
    if (!v || n < 0 || n > ORBV_MAX_FEATURES || !n_bow || !n_fv || !fv_off) return ORBX_ERR_ARG;
    if (n == 0 || v->n_words == 0) { *n_bow = 0; *n_fv = 0; fv_off[0] = 0; return ORBX_OK; }
    if (!desc || !bow_id || !bow_val || !fv_node || !fv_feat) return ORBX_ERR_ARG;
    if (hipSetDevice(v->device) != hipSuccess) return ORBX_ERR_DEVICE;
    // one allocation: desc | bow_val (8-aligned) | bow_id | fv_node | fv_feat | fv_off | counts
    const size_t N = (size_t)n;
    const size_t o_val = N * 32, o_id = o_val + N * 8, o_fn = o_id + N * 4, o_ff = o_fn + N * 4, o_fo = o_ff + N * 4,
                 o_cnt = o_fo + (N + 1) * 4, total = o_cnt + 8;
    uint8_t* d = nullptr;
    int rc = ORBX_ERR_DEVICE;
    if (hipMalloc(&d, total) == hipSuccess && hipMemcpy(d, desc, N * 32, hipMemcpyHostToDevice) == hipSuccess) {
        int32_t* cnt = (int32_t*)(d + o_cnt);
        rc = orbv_transform_batch_device(const_cast<orbv_vocabulary*>(v), d, nullptr, 1, n, levelsup, (uint32_t*)(d + o_id), (double*)(d + o_val),
                                         cnt, (uint32_t*)(d + o_fn), (int32_t*)(d + o_fo), (uint32_t*)(d + o_ff), cnt + 1, nullptr);
        int32_t hc[2] = {0, 0};
        if (rc == ORBX_OK && hipMemcpy(hc, cnt, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = ORBX_ERR_DEVICE;
        if (rc == ORBX_OK) {
            *n_bow = hc[0];
            *n_fv = hc[1];
            if ((hc[0] && (hipMemcpy(bow_id, d + o_id, (size_t)hc[0] * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(bow_val, d + o_val, (size_t)hc[0] * 8, hipMemcpyDeviceToHost) != hipSuccess)) ||
                (hc[1] && hipMemcpy(fv_node, d + o_fn, (size_t)hc[1] * 4, hipMemcpyDeviceToHost) != hipSuccess) ||
                hipMemcpy(fv_off, d + o_fo, ((size_t)hc[1] + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(fv_feat, d + o_ff, N * 4, hipMemcpyDeviceToHost) != hipSuccess)
                rc = ORBX_ERR_DEVICE;
        }
    }
    if (d) (void)hipFree(d);
    return rc;
}

// DBoW2/ScoringObject.cpp: the six merge walks (the reference's lower_bound jumps only skip keys that cannot match)
double orbv_score(const orbv_vocabulary* v, const uint32_t* id1, const double* val1, int n1, const uint32_t* id2, const double* val2, int n2) {
    if (!v) return 0.0;
    const int sc = v->scoring;
    const double log_eps = log(DBL_EPSILON);
    double s = 0;
    int i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (id1[i] == id2[j]) {
            const double a = val1[i], b = val2[j];
            if (sc == ORBV_L1_NORM) s += fabs(a - b) - fabs(a) - fabs(b);
            else if (sc == ORBV_L2_NORM || sc == ORBV_DOT_PRODUCT) s += a * b;
            else if (sc == ORBV_CHI_SQUARE) { if (a + b != 0.0) s += a * b / (a + b); }
            else if (sc == ORBV_KL) { if (a != 0 && b != 0) s += a * log(a / b); }
            else s += sqrt(a * b);
            i++; j++;
        } else if (id1[i] < id2[j]) {
            if (sc == ORBV_KL) s += val1[i] * (log(val1[i]) - log_eps);
            i++;
        } else {
            j++;
        }
    }
    if (sc == ORBV_L1_NORM) return -s / 2.0;
    if (sc == ORBV_L2_NORM) return s >= 1 ? 1.0 : 1.0 - sqrt(1.0 - s);
    if (sc == ORBV_CHI_SQUARE) return 2. * s;
    if (sc == ORBV_KL) {
        for (; i < n1; i++) if (val1[i] != 0) s += val1[i] * (log(val1[i]) - log_eps);
        return s;
    }
    return s;
}

}  // extern "C"
