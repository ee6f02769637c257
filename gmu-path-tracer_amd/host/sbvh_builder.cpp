// See sbvh_builder.hpp.  Rules followed (reference file:line, relative to /root/reference):
//   node decision        Source/Nvidia-SBVH/SplitBVHBuilder.cpp:123-180
//   object split sweep   :195-241, partition :245-254
//   spatial split bins   :267-346, partition + unsplit/duplicate decision :350-448
//   reference clipping   :452-490
//   leaf emission        :184-191 (pops the reference stack, right subtree is built first :176-177)
//   AABB semantics       Include/Nvidia-SBVH/BVHNode.h:41-64 (grow(AABB) = grow(min), grow(max): growing by an
//                        EMPTY box therefore yields a huge box; kept, because it decides which spatial splits
//                        the reference ever takes)
//   flatten              Source/BVHWrapper.cpp:56-95
#include "sbvh_builder.hpp"
#include <atomic>
#include <future>
#include <thread>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cstring>
#include <stdexcept>
#include <utility>

namespace gmupt {

static constexpr int kBins = 32; // SplitBVHBuilder.h:40

Aabb::Aabb() { for (int i = 0; i < 3; i++) { mn[i] = FLT_MAX; mx[i] = -FLT_MAX; } }
void Aabb::grow(const float* p) { for (int i = 0; i < 3; i++) { mn[i] = mn[i] < p[i] ? mn[i] : p[i]; mx[i] = mx[i] > p[i] ? mx[i] : p[i]; } }
void Aabb::grow(const Aabb& o) { float a[3] = { o.mn[0], o.mn[1], o.mn[2] }, b[3] = { o.mx[0], o.mx[1], o.mx[2] }; grow(a); grow(b); }
void Aabb::clip(const Aabb& o) { for (int i = 0; i < 3; i++) { mn[i] = mn[i] > o.mn[i] ? mn[i] : o.mn[i]; mx[i] = mx[i] < o.mx[i] ? mx[i] : o.mx[i]; } }
bool Aabb::valid() const { return mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2]; }
float Aabb::area() const
{
    if (!valid()) return 0.0f;
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return (dx * dy + dy * dz + dz * dx) * 2.0f;
}

static inline float fmin2(float a, float b) { return (a > b) ? b : a; } // linear_math.h:44
static inline float fmin3(float a, float b, float c) { return fmin2(fmin2(a, b), c); }
static inline int toInt(float f) { return (f != f || f >= 2147483648.0f || f < -2147483648.0f) ? INT_MIN : (int)f; } // cvttss2si
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }

SbvhBuilder::SbvhBuilder(const float* vertices, uint32_t numVertices, const int32_t* indices, uint32_t numTriangles,
                         const gmupt_sbvh_params& params)
    : mVerts(vertices), mIdx(indices), mNumTris(numTriangles), mP(params)
{
    for (uint32_t i = 0; i < numTriangles * 3u; i++)
        if (indices[i] < 0 || (uint32_t)indices[i] >= numVertices) throw std::invalid_argument("sbvh: vertex index out of range");
}

namespace {
std::atomic<int> gForkedBuilds{ 0 };                 // subtree builds running on their own thread, over all builders of the process
constexpr int kForkMinRefs = 4096;                   // both children must be at least this large to be worth a thread
int forkLimit() { const unsigned hc = std::thread::hardware_concurrency(); return hc > 1 ? (int)(hc < 32 ? hc : 32) - 1 : 0; }
}

SbvhBuilder::SbvhBuilder(SbvhBuilder& parent, int numRef)
    : mVerts(parent.mVerts), mIdx(parent.mIdx), mNumTris(parent.mNumTris), mP(parent.mP)
{
    mMinOverlap = parent.mMinOverlap;
    mStack.assign(parent.mStack.end() - numRef, parent.mStack.end());
    parent.mStack.resize(parent.mStack.size() - (size_t)numRef);
    mRight.assign((size_t)std::max<int>(numRef, kBins) - 1, Aabb());
}

int32_t SbvhBuilder::absorb(const SbvhBuilder& sub, int32_t subRoot)
{
    const int32_t nodeOffset = (int32_t)mNodes.size(), refOffset = (int32_t)mRefTriangles.size();
    for (SbvhNode n : sub.mNodes) {
        if (n.child[0] < 0) { n.lo += refOffset; n.hi += refOffset; }
        else { n.child[0] += nodeOffset; n.child[1] += nodeOffset; }
        mNodes.push_back(n);
    }
    mRefTriangles.insert(mRefTriangles.end(), sub.mRefTriangles.begin(), sub.mRefTriangles.end());
    mNumDuplicates += sub.mNumDuplicates;
    if (sub.mDepth > mDepth) mDepth = sub.mDepth;
    return subRoot + nodeOffset;
}

void SbvhBuilder::build()
{
    Spec rootSpec; rootSpec.numRef = (int32_t)mNumTris;
    mStack.resize(mNumTris);
    for (uint32_t i = 0; i < mNumTris; i++) {
        mStack[i].tri = (int32_t)i;
        mStack[i].b = Aabb();
        for (int j = 0; j < 3; j++) mStack[i].b.grow(mVerts + 3 * (size_t)mIdx[3 * (size_t)i + j]);
        rootSpec.b.grow(mStack[i].b);
    }
    mMinOverlap = rootSpec.b.area() * mP.split_alpha;
    mRight.assign((size_t)std::max<int>(rootSpec.numRef, kBins) - 1, Aabb());
    mNumDuplicates = 0; mDepth = 0;
    mNodes.clear(); mRefTriangles.clear();
    mNodes.reserve((size_t)mNumTris * 2 + 16);
    mRefTriangles.reserve((size_t)mNumTris + mNumTris / 4);
    if (mNumTris == 0) { // degenerate: single empty leaf
        SbvhNode n; n.bounds = Aabb(); n.child[0] = n.child[1] = -1; n.lo = n.hi = 0;
        mNodes.push_back(n); mRoot = 0; return;
    }
    mRoot = buildNode(rootSpec, 0);
}

int32_t SbvhBuilder::makeLeaf(const Spec& spec)
{
    for (int i = 0; i < spec.numRef; i++) { mRefTriangles.push_back(mStack.back().tri); mStack.pop_back(); }
    SbvhNode n; n.bounds = spec.b; n.child[0] = n.child[1] = -1;
    n.lo = (int32_t)mRefTriangles.size() - spec.numRef; n.hi = (int32_t)mRefTriangles.size();
    mNodes.push_back(n);
    return (int32_t)mNodes.size() - 1;
}

int32_t SbvhBuilder::buildNode(const Spec& spec, int level)
{
    if ((uint32_t)level > mDepth) mDepth = (uint32_t)level;
    if (spec.numRef <= mP.min_leaf_size || level >= mP.max_depth) return makeLeaf(spec);

    const float area = spec.b.area();
    const float leafSAH = area * triCost(spec.numRef);
    const float nodeSAH = area * (2.0f * mP.node_cost);
    ObjSplit object = findObjectSplit(spec, nodeSAH);

    SpaSplit spatial; spatial.sah = FLT_MAX; spatial.dim = 0; spatial.pos = 0.0f;
    if (level < mP.max_spatial_depth) {
        Aabb overlap = object.lb;
        overlap.clip(object.rb);
        if (overlap.area() >= mMinOverlap) spatial = findSpatialSplit(spec, nodeSAH);
    }

    const float minSAH = fmin3(leafSAH, object.sah, spatial.sah);
    if (minSAH == leafSAH && spec.numRef <= mP.max_leaf_size) return makeLeaf(spec);

    Spec left, right; left.numRef = right.numRef = 0;
    if (minSAH == spatial.sah) doSpatialSplit(left, right, spec, spatial);
    if (!left.numRef || !right.numRef) doObjectSplit(left, right, spec, object);

    mNumDuplicates += (uint32_t)(left.numRef + right.numRef - spec.numRef);
    // the right child owns the top of the reference stack, so it is built first -- or, when both children are large and a core is free,
    // on another thread with its own copy of those references (same tree: a node only ever looks at its own references)
    int32_t rightNode, leftNode;
    bool forked = false;
    if (left.numRef >= kForkMinRefs && right.numRef >= kForkMinRefs) {
        const int limit = forkLimit();
        if (gForkedBuilds.fetch_add(1) < limit) forked = true; else gForkedBuilds.fetch_sub(1);
    }
    if (forked) {
        SbvhBuilder sub(*this, right.numRef);
        std::future<int32_t> rightDone = std::async(std::launch::async, [&sub, right, level]() { return sub.buildNode(right, level + 1); });
        try { leftNode = buildNode(left, level + 1); }
        catch (...) { rightDone.wait(); gForkedBuilds.fetch_sub(1); throw; }
        int32_t subRoot;
        try { subRoot = rightDone.get(); } catch (...) { gForkedBuilds.fetch_sub(1); throw; }
        gForkedBuilds.fetch_sub(1);
        rightNode = absorb(sub, subRoot);
    } else {
        rightNode = buildNode(right, level + 1);
        leftNode = buildNode(left, level + 1);
    }
    SbvhNode n; n.bounds = spec.b; n.child[0] = leftNode; n.child[1] = rightNode; n.lo = n.hi = 0;
    mNodes.push_back(n);
    return (int32_t)mNodes.size() - 1;
}

void SbvhBuilder::sortTail(int numRef, int dim)
{
    // strict total order inside a node (centroid*2 along dim, then triangle id): the result does not depend on the
    // sort algorithm (SplitBVHBuilder.cpp:103-112)
    std::sort(mStack.end() - numRef, mStack.end(), [dim](const Ref& a, const Ref& b) {
        const float ca = a.b.mn[dim] + a.b.mx[dim], cb = b.b.mn[dim] + b.b.mx[dim];
        return (ca < cb) || (ca == cb && a.tri < b.tri);
    });
}

SbvhBuilder::ObjSplit SbvhBuilder::findObjectSplit(const Spec& spec, float nodeSAH)
{
    ObjSplit best; best.sah = FLT_MAX; best.dim = 0; best.numLeft = 0;
    const int n = spec.numRef;
    for (int dim = 0; dim < 3; dim++) {
        sortTail(n, dim);
        const Ref* refs = mStack.data() + (mStack.size() - (size_t)n);
        Aabb rb;
        for (int i = n - 1; i > 0; i--) { rb.grow(refs[i].b); mRight[(size_t)i - 1] = rb; }
        Aabb lb;
        for (int i = 1; i < n; i++) {
            lb.grow(refs[i - 1].b);
            const float sah = nodeSAH + lb.area() * triCost(i) + mRight[(size_t)i - 1].area() * triCost(n - i);
            if (sah < best.sah) { best.sah = sah; best.dim = dim; best.numLeft = i; best.lb = lb; best.rb = mRight[(size_t)i - 1]; }
        }
    }
    return best;
}

void SbvhBuilder::doObjectSplit(Spec& l, Spec& r, const Spec& spec, const ObjSplit& s)
{
    sortTail(spec.numRef, s.dim);
    l.numRef = s.numLeft; l.b = s.lb;
    r.numRef = spec.numRef - s.numLeft; r.b = s.rb;
}

SbvhBuilder::SpaSplit SbvhBuilder::findSpatialSplit(const Spec& spec, float nodeSAH)
{
    struct Bin { Aabb b; int enter, exit; };
    Bin bins[3][kBins];
    float origin[3], binSize[3], invBin[3];
    for (int d = 0; d < 3; d++) {
        origin[d] = spec.b.mn[d];
        binSize[d] = (spec.b.mx[d] - origin[d]) * (1.0f / (float)kBins);
        invBin[d] = 1.0f / binSize[d];
        for (int i = 0; i < kBins; i++) { bins[d][i].b = Aabb(); bins[d][i].enter = 0; bins[d][i].exit = 0; }
    }
    for (size_t ri = mStack.size() - (size_t)spec.numRef; ri < mStack.size(); ri++) {
        const Ref ref = mStack[ri];
        int first[3], last[3];
        for (int d = 0; d < 3; d++) first[d] = clampi(toInt((ref.b.mn[d] - origin[d]) * invBin[d]), 0, kBins - 1);
        for (int d = 0; d < 3; d++) last[d] = clampi(toInt((ref.b.mx[d] - origin[d]) * invBin[d]), first[d], kBins - 1);
        for (int d = 0; d < 3; d++) {
            Ref cur = ref;
            for (int i = first[d]; i < last[d]; i++) {
                Ref lr, rr;
                splitRef(lr, rr, cur, d, origin[d] + binSize[d] * (float)(i + 1));
                bins[d][i].b.grow(lr.b);
                cur = rr;
            }
            bins[d][last[d]].b.grow(cur.b);
            bins[d][first[d]].enter++;
            bins[d][last[d]].exit++;
        }
    }
    SpaSplit best; best.sah = FLT_MAX; best.dim = 0; best.pos = 0.0f;
    for (int d = 0; d < 3; d++) {
        Aabb rb;
        for (int i = kBins - 1; i > 0; i--) { rb.grow(bins[d][i].b); mRight[(size_t)i - 1] = rb; }
        Aabb lb; int leftNum = 0, rightNum = spec.numRef;
        for (int i = 1; i < kBins; i++) {
            lb.grow(bins[d][i - 1].b);
            leftNum += bins[d][i - 1].enter;
            rightNum -= bins[d][i - 1].exit;
            const float sah = nodeSAH + lb.area() * triCost(leftNum) + mRight[(size_t)i - 1].area() * triCost(rightNum);
            if (sah < best.sah) { best.sah = sah; best.dim = d; best.pos = origin[d] + binSize[d] * (float)i; }
        }
    }
    return best;
}

void SbvhBuilder::doSpatialSplit(Spec& left, Spec& right, const Spec& spec, const SpaSplit& s)
{
    std::vector<Ref>& refs = mStack;
    const int leftStart = (int)refs.size() - spec.numRef;
    int leftEnd = leftStart, rightStart = (int)refs.size();
    left.b = Aabb(); right.b = Aabb();
    for (int i = leftEnd; i < rightStart; i++) {
        if (refs[(size_t)i].b.mx[s.dim] <= s.pos) { left.b.grow(refs[(size_t)i].b); std::swap(refs[(size_t)i], refs[(size_t)leftEnd++]); }
        else if (refs[(size_t)i].b.mn[s.dim] >= s.pos) { right.b.grow(refs[(size_t)i].b); --rightStart; std::swap(refs[(size_t)i], refs[(size_t)rightStart]); i--; }
    }
    while (leftEnd < rightStart) {
        Ref lref, rref;
        splitRef(lref, rref, refs[(size_t)leftEnd], s.dim, s.pos);
        Aabb lub = left.b, rub = right.b, ldb = left.b, rdb = right.b;
        lub.grow(refs[(size_t)leftEnd].b); rub.grow(refs[(size_t)leftEnd].b);
        ldb.grow(lref.b); rdb.grow(rref.b);
        const float lac = triCost(leftEnd - leftStart), rac = triCost((int)refs.size() - rightStart);
        const float lbc = triCost(leftEnd - leftStart + 1), rbc = triCost((int)refs.size() - rightStart + 1);
        const float unsplitLeft = lub.area() * lbc + right.b.area() * rac;
        const float unsplitRight = left.b.area() * lac + rub.area() * rbc;
        const float duplicate = ldb.area() * lbc + rdb.area() * rbc;
        const float m = fmin3(unsplitLeft, unsplitRight, duplicate);
        if (m == unsplitLeft) { left.b = lub; leftEnd++; }
        else if (m == unsplitRight) { right.b = rub; --rightStart; std::swap(refs[(size_t)leftEnd], refs[(size_t)rightStart]); }
        else { left.b = ldb; right.b = rdb; refs[(size_t)leftEnd++] = lref; refs.push_back(rref); }
    }
    left.numRef = leftEnd - leftStart;
    right.numRef = (int)refs.size() - rightStart;
}

void SbvhBuilder::splitRef(Ref& l, Ref& r, const Ref& ref, int dim, float pos) const
{
    l.tri = r.tri = ref.tri;
    l.b = Aabb(); r.b = Aabb();
    const int32_t* ind = mIdx + 3 * (size_t)ref.tri;
    const float* v1 = mVerts + 3 * (size_t)ind[2];
    for (int i = 0; i < 3; i++) {
        const float* v0 = v1;
        v1 = mVerts + 3 * (size_t)ind[i];
        const float v0p = v0[dim], v1p = v1[dim];
        if (v0p <= pos) l.b.grow(v0);
        if (v0p >= pos) r.b.grow(v0);
        if ((v0p < pos && v1p > pos) || (v0p > pos && v1p < pos)) {
            const float t = clampf((pos - v0p) / (v1p - v0p), 0.0f, 1.0f);
            const float p[3] = { v0[0] * (1.0f - t) + v1[0] * t, v0[1] * (1.0f - t) + v1[1] * t, v0[2] * (1.0f - t) + v1[2] * t }; // Sort.h:49 lerp
            l.b.grow(p); r.b.grow(p);
        }
    }
    l.b.mx[dim] = pos; r.b.mn[dim] = pos;
    l.b.clip(ref.b); r.b.clip(ref.b);
}

float SbvhBuilder::sah() const
{
    // BVHNode::computeSubtreeProbabilities (Source/Nvidia-SBVH/BVHNode.cpp:65-79): pre-order, child 0 first
    float total = 0.0f;
    struct Item { int32_t node; float prob; };
    std::vector<Item> st; st.push_back({ mRoot, 1.0f });
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        const SbvhNode& n = mNodes[(size_t)it.node];
        const bool leaf = n.child[0] < 0;
        const float cost = (leaf ? 0.0f : 2.0f * mP.node_cost) + (leaf ? triCost(n.hi - n.lo) : 0.0f);
        total += it.prob * cost;
        if (!leaf) {
            const float pa = n.bounds.area();
            st.push_back({ n.child[1], it.prob * mNodes[(size_t)n.child[1]].bounds.area() / pa });
            st.push_back({ n.child[0], it.prob * mNodes[(size_t)n.child[0]].bounds.area() / pa });
        }
    }
    return total;
}

void SbvhBuilder::flatten(const uint32_t* vertexMaterial, gmupt_bvh_node* outNodes, gmupt_triangle* outTris, int32_t* outRefTri) const
{
    std::vector<std::pair<int32_t, uint32_t>> st; st.push_back({ mRoot, 0u });
    uint32_t nodeIndex = 0, triCount = 0;
    while (!st.empty()) {
        const auto [ni, cur] = st.back(); st.pop_back();
        const SbvhNode& n = mNodes[(size_t)ni];
        gmupt_bvh_node& o = outNodes[cur];
        std::memset(&o, 0, sizeof(o));
        for (int k = 0; k < 3; k++) { o.min[k] = n.bounds.mn[k]; o.max[k] = n.bounds.mx[k]; }
        if (n.child[0] < 0) {
            o.left = (int32_t)triCount; o.right = (int32_t)triCount + (n.hi - n.lo); o.isLeaf = 1;
            for (int32_t i = n.lo; i < n.hi; i++) {
                const int32_t t = mRefTriangles[(size_t)i];
                const int32_t* ind = mIdx + 3 * (size_t)t;
                if (outTris) {
                    gmupt_triangle& T = outTris[triCount];
                    T.v[0] = ind[0]; T.v[1] = ind[1]; T.v[2] = ind[2];
                    T.materialID = vertexMaterial ? vertexMaterial[ind[0]] : 0u; // BVHWrapper.cpp:82 (first vertex's material)
                }
                if (outRefTri) outRefTri[triCount] = t;
                triCount++;
            }
        } else {
            nodeIndex += 2;
            st.push_back({ n.child[1], nodeIndex }); o.right = (int32_t)nodeIndex;
            st.push_back({ n.child[0], nodeIndex - 1 }); o.left = (int32_t)nodeIndex - 1;
            o.isLeaf = 0;
        }
    }
}

} // namespace gmupt
