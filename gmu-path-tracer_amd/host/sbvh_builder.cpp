// Split-BVH builder: see sbvh_builder.hpp for the design.  This file states, step by step, which rule of the reference's builder
// (file:line relative to /root/reference) each piece reproduces -- the RULES are the reference's (the tree has to be identical), the
// program is not: pre-sorted reference lists handed down by stable partition instead of per-node sorts, a task pool instead of
// recursion over one shared stack, pointer-linked nodes numbered after the fact.
//
//   root set-up              Source/Nvidia-SBVH/SplitBVHBuilder.cpp:63-83     (buildAll)
//   node decision            :126-188   (decideNode)
//   object-split candidates  :202-242   (sweepAxis / searchObjectSplit)  order of the candidates: axis 0,1,2, split position ascending,
//                                        strict "<" keeps the first minimum; partition :244-253 (applyObjectSplit)
//   reference order          :103-114   total order inside a node: centroid*2 along the axis, then triangle id (sortKey)
//   spatial-split bins       :265-347   (searchSpatialSplit)
//   spatial partition        :349-443   (applySpatialSplit: see the comment there for how the reference's in-place swap loop is restated)
//   reference clipping       :445-486   (chopReference)
//   leaf contents            :190-200   a leaf takes the references in the REVERSE of the order they have at that moment
//   box arithmetic           Include/Nvidia-SBVH/BVHNode.h:41-64, linear_math.h:43-44,120-121: min / max return the SECOND operand on
//                            a tie, so the sign of a zero bound depends on the order of the union -- node boxes are therefore
//                            accumulated in the reference's order wherever they end up in the tree; growing by an EMPTY box grows by
//                            its two corner points and yields a huge box (it decides which spatial splits the reference ever takes)
//   flatten                  Source/BVHWrapper.cpp:56-95
#include "sbvh_builder.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

namespace gmupt {
namespace {

constexpr int kBins = 32;                       // SplitBVHBuilder.h:40
constexpr uint32_t kSpawnMinRefs = 2048;        // a child with fewer references stays on the worker that made it
constexpr uint32_t kFanOutMinRefs = 1u << 19;   // nodes at least this large spread their linear passes over helper threads (GMUPT_BUILD_FANOUT overrides: tests)

// ---------------------------------------------------------------------------------------------- boxes
struct Box {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    static Box empty() { Box b; b.clear(); return b; }
    // min3f / max3f of linear_math.h:120-121: the new point wins a tie (this fixes the sign of a zero bound)
    void addPoint(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = lo[a] < p[a] ? lo[a] : p[a]; hi[a] = hi[a] > p[a] ? hi[a] : p[a]; } }
    void include(const Box& o) { addPoint(o.lo); addPoint(o.hi); }               // AABB::grow(AABB): by its two corner points, empty or not
    void clipTo(const Box& o) { for (int a = 0; a < 3; a++) { lo[a] = lo[a] > o.lo[a] ? lo[a] : o.lo[a]; hi[a] = hi[a] < o.hi[a] ? hi[a] : o.hi[a]; } } // AABB::intersect
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
    float area() const
    {
        if (!valid()) return 0.0f;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx * dy + dy * dz + dz * dx) * 2.0f;
    }
};

inline float least(float a, float b) { return (a > b) ? b : a; }                   // min1f, linear_math.h:44
inline float least(float a, float b, float c) { return least(least(a, b), c); }
inline int truncToInt(float f) { return (f != f || f >= 2147483648.0f || f < -2147483648.0f) ? INT_MIN : (int)f; } // what the reference's (int) cast does on x86 (cvttss2si)
inline int clampInt(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
inline float clampFloat(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; } // linear_math.h:129

// ---------------------------------------------------------------------------------------------- references
struct RefRec { Box box; int32_t tri; uint32_t mark; };  // 32 bytes; mark: scratch of the node that currently owns the reference
static_assert(sizeof(RefRec) == 32, "reference record");
enum : uint32_t { MARK_LEFT = 0, MARK_RIGHT = 1, MARK_REPLACED = 2 };

// Chunked append-only pool: records never move, ids stay valid while other workers append.
class RefPool {
public:
    static constexpr uint32_t kChunkBits = 16, kChunkSize = 1u << kChunkBits, kMaxChunks = 1u << 16;
    RefPool() : mChunks(kMaxChunks) { for (auto& c : mChunks) c.store(nullptr, std::memory_order_relaxed); }
    ~RefPool() { for (auto& c : mChunks) delete[] c.load(std::memory_order_relaxed); }
    RefRec& operator[](uint32_t id) { return mChunks[id >> kChunkBits].load(std::memory_order_relaxed)[id & (kChunkSize - 1)]; }
    const RefRec& operator[](uint32_t id) const { return mChunks[id >> kChunkBits].load(std::memory_order_relaxed)[id & (kChunkSize - 1)]; }
    // reserves `count` consecutive ids (count <= kChunkSize; a block never straddles... it may: ids are only used one at a time)
    uint32_t reserve(uint32_t count)
    {
        const uint64_t first = mNext.fetch_add(count);
        if (first + count > (uint64_t)kMaxChunks * kChunkSize) throw std::length_error("sbvh: reference pool exhausted");
        for (uint64_t c = first >> kChunkBits; c <= (first + count - 1) >> kChunkBits; c++) ensure((uint32_t)c);
        return (uint32_t)first;
    }
    uint64_t size() const { return mNext.load(); }
private:
    void ensure(uint32_t c)
    {
        if (mChunks[c].load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> g(mGrow);
        if (!mChunks[c].load(std::memory_order_relaxed)) mChunks[c].store(new RefRec[kChunkSize], std::memory_order_release);
    }
    std::vector<std::atomic<RefRec*>> mChunks;
    std::atomic<uint64_t> mNext{ 0 };
    std::mutex mGrow;
};

// The reference's comparator (SplitBVHBuilder.cpp:103-112) as one integer key: centroid*2 along the axis, ties by triangle id.
// -0 and +0 compare equal as floats, so the sum is canonicalised to +0 before its bits are made order-preserving.
inline uint32_t orderedBits(float f)
{
    f += 0.0f;
    uint32_t u; std::memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline uint64_t sortKey(const RefRec& r, int axis) { return ((uint64_t)orderedBits(r.box.lo[axis] + r.box.hi[axis]) << 32) | (uint32_t)r.tri; }

struct KeyedRef { uint64_t key; uint32_t ref; };

// ---------------------------------------------------------------------------------------------- helper threads for big nodes
// Runs fn(part, begin, end) over `parts` contiguous slices of [0, n) on their own threads (the caller takes slice 0).
template <typename Fn>
void forSlices(size_t n, int parts, Fn fn)
{
    if (parts <= 1 || n < (size_t)parts) { fn(0, (size_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)parts - 1);
    for (int p = 1; p < parts; p++) th.emplace_back([=]() { fn(p, n * (size_t)p / (size_t)parts, n * (size_t)(p + 1) / (size_t)parts); });
    fn(0, (size_t)0, n / (size_t)parts);
    for (auto& t : th) t.join();
}

// ascending by key; keys are unique inside a node (a triangle occurs once per node)
void sortKeyed(std::vector<KeyedRef>& a, int threads, size_t parallelFrom = (size_t)1 << 16)
{
    auto less = [](const KeyedRef& x, const KeyedRef& y) { return x.key < y.key; };
    const size_t n = a.size();
    if (threads <= 1 || n < parallelFrom || n < (size_t)threads * 2) { std::sort(a.begin(), a.end(), less); return; }
    int parts = 1; while (parts * 2 <= threads) parts *= 2;
    std::vector<size_t> cut((size_t)parts + 1);
    for (int p = 0; p <= parts; p++) cut[(size_t)p] = n * (size_t)p / (size_t)parts;
    forSlices((size_t)parts, parts, [&](int, size_t b, size_t e) { for (size_t p = b; p < e; p++) std::sort(a.begin() + (long)cut[p], a.begin() + (long)cut[p + 1], less); });
    std::vector<KeyedRef> tmp(n);
    std::vector<KeyedRef>* src = &a; std::vector<KeyedRef>* dst = &tmp;
    for (int width = 1; width < parts; width *= 2) {          // merge runs of `width` slices pairwise, all pairs of a round in parallel
        const int pairs = parts / (2 * width);
        forSlices((size_t)pairs, pairs, [&](int, size_t b, size_t e) {
            for (size_t q = b; q < e; q++) {
                const size_t lo = cut[q * 2 * (size_t)width], mid = cut[q * 2 * (size_t)width + (size_t)width], hi = cut[(q + 1) * 2 * (size_t)width];
                std::merge(src->begin() + (long)lo, src->begin() + (long)mid, src->begin() + (long)mid, src->begin() + (long)hi, dst->begin() + (long)lo, less);
            }
        });
        std::swap(src, dst);
    }
    if (src != &a) a.swap(tmp);
}

// ---------------------------------------------------------------------------------------------- tree
struct Node {
    Box bounds;
    Node* child[2];          // both null: leaf
    const int32_t* leafTris; // leaf: triangle ids in the order the reference's leaf holds them
    uint32_t leafCount;
};

// The three sorted lists of a node are slices [offset, offset + n) of the three arrays of a ListBlock.  The root's block is allocated
// once; an object split (and a spatial split that duplicates nothing) partitions the slices IN PLACE, so its children are sub-slices
// of the same block -- no allocation, no page faults; only a spatial split that creates references needs a (larger) block of its own.
struct ListBlock {
    std::unique_ptr<uint32_t[]> mem; size_t cap;
    explicit ListBlock(size_t n) : mem(new uint32_t[3 * (n ? n : 1)]), cap(n ? n : 1) {}
    uint32_t* axis(int a) { return mem.get() + (size_t)a * cap; }
};

struct Task {
    Node** slot;                     // where the finished node is linked (parent's child pointer, or the root pointer)
    Box bounds;
    int level;
    uint32_t n;
    std::shared_ptr<ListBlock> block; size_t offset;   // the node's references, sorted along x, y, z by sortKey: block->axis(a) + offset
    uint32_t* list(int a) const { return block->axis(a) + offset; }
    // The order in which the references "arrive" (what the reference's stack holds when the node is created) only matters for nodes that
    // become a leaf before they sort: -1 = `arrival` holds it explicitly; 0..2 = it is list(arrivalAxis) (child of an object split).
    int arrivalAxis;
    std::vector<uint32_t> arrival;
};

struct ObjectPlan { float sah; int axis; uint32_t numLeft; Box lb, rb; };
struct SpatialPlan { float sah; int axis; float pos; };

struct Worker {
    std::deque<Node> nodes;                              // stable addresses
    std::vector<std::unique_ptr<int32_t[]>> triBlocks;   // leaf triangle lists
    size_t triUsed = 0, triCap = 0;
    uint32_t refNext = 0, refEnd = 0;                    // private block of fresh reference ids
    uint64_t leafRefs = 0, duplicates = 0;
    uint32_t maxLevel = 0;
    std::vector<float> areas;                            // scratch of the sweeps
    std::vector<uint32_t> moved;                         // scratch of the in-place partitions
    // scratch of the spatial partition (kept between nodes: a build makes thousands of them, and fresh large allocations cost page faults)
    std::vector<uint32_t> leftBlock, rightFront, rightTail, waiting, replaced;
    std::vector<std::pair<uint32_t, Box>> madeLeft, madeRight;
    std::vector<KeyedRef> newLeft, newRight;
    int32_t* allocTris(size_t count)
    {
        if (triUsed + count > triCap) {
            triCap = std::max<size_t>(count, (size_t)1 << 18);
            triBlocks.emplace_back(new int32_t[triCap]);
            triUsed = 0;
        }
        int32_t* p = triBlocks.back().get() + triUsed;
        triUsed += count;
        return p;
    }
};

} // namespace

// ---------------------------------------------------------------------------------------------- builder state
struct SbvhBuilder::Impl {
    const float* verts; const int32_t* idx; uint32_t numTris; gmupt_sbvh_params prm;
    RefPool refs;
    float minOverlap = 0.0f;
    Node* root = nullptr;
    std::vector<std::unique_ptr<Worker>> workers;
    int threads = 1;
    bool built = false, verbose = false;
    // task pool
    std::mutex qMutex; std::condition_variable qCond;
    std::vector<Task*> queue;
    int pending = 0;                 // tasks queued or being worked on (under qMutex)
    std::atomic<int> failAfterTasks{ -1 };   // test hook (GMUPT_SBVH_FAIL_AFTER): the n-th task taken from the queue throws
    bool stop = false;               // a worker failed: queued tasks are dropped, nobody waits any more (under qMutex)
    std::exception_ptr failure;
    // totals after build
    uint32_t totalNodes = 0, totalRefs = 0, totalDup = 0, maxLevel = 0;

    float triCost(uint32_t n) const { return (float)(int)n * prm.tri_cost; }
    uint32_t fanOutFrom = kFanOutMinRefs;
    int helpersFor(uint32_t n) const { return n >= fanOutFrom ? threads : 1; }

    uint32_t newRef(Worker& w, const Box& b, int32_t tri)
    {
        if (w.refNext == w.refEnd) { w.refNext = refs.reserve(256); w.refEnd = w.refNext + 256; }
        const uint32_t id = w.refNext++;
        RefRec& r = refs[id]; r.box = b; r.tri = tri; r.mark = 0;
        return id;
    }

    void buildAll();
    void workerLoop(int wi);
    void push(Task* t);
    void finishSubtree(Task* first, Worker& w);
    void decideNode(Task& t, Worker& w, Task*& left, Task*& right);
    void makeLeaf(Node* node, const uint32_t* listInArrivalOrder, uint32_t n, Worker& w);
    void sweepAxis(const Task& t, int axis, float nodeSAH, Worker& w, ObjectPlan& best) const;
    ObjectPlan searchObjectSplit(const Task& t, float nodeSAH, Worker& w) const;
    SpatialPlan searchSpatialSplit(const Task& t, float nodeSAH) const;
    void chopReference(Box& l, Box& r, const RefRec& ref, int axis, float pos) const;
    void applyObjectSplit(Task& t, const ObjectPlan& plan, Worker& w, Task*& left, Task*& right);
    bool applySpatialSplit(Task& t, const SpatialPlan& plan, Worker& w, Task*& left, Task*& right);
    void partitionInPlace(const Task& t, int skipAxis, uint32_t numLeft, Worker& w);
};

SbvhBuilder::SbvhBuilder(const float* vertices, uint32_t numVertices, const int32_t* indices, uint32_t numTriangles, const gmupt_sbvh_params& params)
    : m(new Impl())
{
    m->verts = vertices; m->idx = indices; m->numTris = numTriangles; m->prm = params;
    for (size_t i = 0; i < (size_t)numTriangles * 3u; i++)
        if (indices[i] < 0 || (uint32_t)indices[i] >= numVertices) throw std::invalid_argument("sbvh: vertex index out of range");
    const char* env = std::getenv("GMUPT_BUILD_THREADS");
    const unsigned hc = std::thread::hardware_concurrency();
    int t = env ? std::atoi(env) : (int)(hc ? (hc < 16 ? hc : 16) : 1);   // 16 = the CPU share of one GPU on the target boxes
    m->threads = t < 1 ? 1 : (t > 64 ? 64 : t);
    if (const char* f = std::getenv("GMUPT_BUILD_FANOUT")) m->fanOutFrom = (uint32_t)std::max(2, std::atoi(f));
    if (const char* f = std::getenv("GMUPT_SBVH_FAIL_AFTER")) m->failAfterTasks = std::atoi(f);   // test hook: the build must fail, not hang
}

SbvhBuilder::~SbvhBuilder() = default;
uint32_t SbvhBuilder::numNodes() const { return m->totalNodes; }
uint32_t SbvhBuilder::numReferences() const { return m->totalRefs; }
uint32_t SbvhBuilder::numDuplicates() const { return m->totalDup; }
uint32_t SbvhBuilder::depth() const { return m->maxLevel; }

void SbvhBuilder::build()
{
    if (m->built) throw std::logic_error("sbvh: build() called twice");
    m->built = true;
    m->buildAll();
}

// ---------------------------------------------------------------------------------------------- root set-up and the task pool
void SbvhBuilder::Impl::buildAll()
{
    verbose = std::getenv("GMUPT_BUILD_VERBOSE") != nullptr;   // GMUPT_BUILD_VERBOSE=1: phase times on stderr
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (verbose) std::fprintf(stderr, "sbvh: %-28s %8.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); };
    for (int w = 0; w < threads; w++) workers.emplace_back(new Worker());
    const uint32_t n = numTris;
    Task* rootTask = new Task();
    rootTask->slot = &root; rootTask->level = 0; rootTask->n = n; rootTask->arrivalAxis = -1;
    rootTask->bounds.clear();
    if (n) {
        const uint32_t base = refs.reserve(n);   // the root's references: id == triangle id
        (void)base;
        forSlices(n, helpersFor(n), [&](int, size_t b, size_t e) {
            for (size_t i = b; i < e; i++) {
                RefRec& r = refs[(uint32_t)i];
                r.tri = (int32_t)i; r.mark = 0; r.box.clear();
                for (int j = 0; j < 3; j++) r.box.addPoint(verts + 3 * (size_t)idx[3 * i + (size_t)j]);   // SplitBVHBuilder.cpp:69-76
            }
        });
        for (uint32_t i = 0; i < n; i++) rootTask->bounds.include(refs[i].box);                           // :78, in triangle order
    }
    lap("reference boxes, root box");
    minOverlap = rootTask->bounds.area() * prm.split_alpha;                                               // :83
    rootTask->block = std::make_shared<ListBlock>(n); rootTask->offset = 0;
    if (n <= (uint32_t)std::max(prm.min_leaf_size, 0)) {          // the root never sorts: its references arrive in triangle order
        rootTask->arrival.resize(n);
        for (uint32_t i = 0; i < n; i++) rootTask->arrival[i] = i;
        for (int a = 0; a < 3; a++) for (uint32_t i = 0; i < n; i++) rootTask->list(a)[i] = i;
    } else {
        std::vector<KeyedRef> keyed(n);
        for (int a = 0; a < 3; a++) {
            forSlices(n, helpersFor(n), [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) keyed[i] = { sortKey(refs[(uint32_t)i], a), (uint32_t)i }; });
            sortKeyed(keyed, threads, std::min<size_t>(fanOutFrom, (size_t)1 << 16));
            uint32_t* dst = rootTask->list(a);
            for (uint32_t i = 0; i < n; i++) dst[i] = keyed[i].ref;
        }
    }
    lap("three sorted lists");
    queue.push_back(rootTask); pending = 1;
    std::vector<std::thread> pool;
    for (int w = 1; w < threads; w++) pool.emplace_back([this, w]() { workerLoop(w); });
    workerLoop(0);
    for (auto& t : pool) t.join();
    lap("tree");
    if (failure) std::rethrow_exception(failure);
    uint64_t nodes = 0, lrefs = 0, dup = 0;
    for (const auto& wp : workers) { const Worker& w = *wp; nodes += w.nodes.size(); lrefs += w.leafRefs; dup += w.duplicates; if (w.maxLevel > maxLevel) maxLevel = w.maxLevel; }
    if (nodes > 0x7FFFFFFFull || lrefs > 0x7FFFFFFFull) throw std::length_error("sbvh: tree exceeds 2^31 nodes or references");
    totalNodes = (uint32_t)nodes; totalRefs = (uint32_t)lrefs; totalDup = (uint32_t)dup;
}

void SbvhBuilder::Impl::push(Task* t)
{
    {
        std::lock_guard<std::mutex> g(qMutex);
        if (stop) { delete t; return; }     // after a failure nothing new is queued (the tree is abandoned)
        queue.push_back(t); pending++;
    }
    qCond.notify_one();
}

void SbvhBuilder::Impl::workerLoop(int wi)
{
    Worker& w = *workers[(size_t)wi];
    for (;;) {
        Task* t = nullptr;
        {
            std::unique_lock<std::mutex> lk(qMutex);
            qCond.wait(lk, [&]() { return stop || !queue.empty() || pending == 0; });
            if (stop || queue.empty()) return;      // a worker failed, or pending == 0: the tree is complete
            t = queue.back(); queue.pop_back();
        }
        try {
            if (failAfterTasks >= 0 && failAfterTasks-- == 0) { delete t; throw std::runtime_error("sbvh: injected failure (test)"); }
            finishSubtree(t, w);
        }
        catch (...) {
            // the first failure wins; every queued task is dropped and every worker -- waiting now or coming back from its task later --
            // leaves through `stop` (a counter could not say that: workers still inside a task would decrement it below zero)
            { std::lock_guard<std::mutex> g(qMutex); if (!failure) failure = std::current_exception(); for (Task* q : queue) delete q; queue.clear(); stop = true; }
            qCond.notify_all();
            return;
        }
        bool done;
        { std::lock_guard<std::mutex> g(qMutex); done = (--pending == 0); }
        if (done) qCond.notify_all();
    }
}

// One queued task: the node itself, then everything below it that is not worth queueing, on a private stack.
void SbvhBuilder::Impl::finishSubtree(Task* first, Worker& w)
{
    std::vector<Task*> mine; mine.push_back(first);
    while (!mine.empty()) {
        Task* t = mine.back(); mine.pop_back();
        Task *l = nullptr, *r = nullptr;
        try { decideNode(*t, w, l, r); }
        catch (...) { delete t; delete l; delete r; for (Task* q : mine) delete q; throw; }
        delete t;
        for (Task* c : { r, l }) {      // the left child is worked on first (it stays on top of the private stack)
            if (!c) continue;
            if (threads > 1 && c->n >= kSpawnMinRefs && c != l) push(c); else mine.push_back(c);
        }
    }
}

// ---------------------------------------------------------------------------------------------- leaves
void SbvhBuilder::Impl::makeLeaf(Node* node, const uint32_t* list, uint32_t n, Worker& w)
{
    // createLeaf pops the node's references off the END of the stack (SplitBVHBuilder.cpp:190-200): reverse arrival order
    int32_t* tris = n ? w.allocTris(n) : nullptr;
    for (uint32_t i = 0; i < n; i++) tris[i] = refs[list[n - 1 - i]].tri;
    node->child[0] = node->child[1] = nullptr; node->leafTris = tris; node->leafCount = n;
    w.leafRefs += n;
}

// ---------------------------------------------------------------------------------------------- one node
void SbvhBuilder::Impl::decideNode(Task& t, Worker& w, Task*& left, Task*& right)
{
    w.nodes.emplace_back();
    Node* node = &w.nodes.back();
    node->bounds = t.bounds; node->child[0] = node->child[1] = nullptr; node->leafTris = nullptr; node->leafCount = 0;
    *t.slot = node;
    if ((uint32_t)t.level > w.maxLevel) w.maxLevel = (uint32_t)t.level;

    // small enough, or too deep: a leaf of the references as they arrived (SplitBVHBuilder.cpp:127-128)
    if ((int64_t)t.n <= (int64_t)prm.min_leaf_size || t.level >= prm.max_depth) {
        makeLeaf(node, t.arrivalAxis < 0 ? t.arrival.data() : t.list(t.arrivalAxis), t.n, w);
        return;
    }

    const bool big = verbose && t.n >= (1u << 20);
    const auto tb0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count(); };
    const float area = t.bounds.area();
    const float leafSAH = area * triCost(t.n);                     // :148
    const float nodeSAH = area * (2.0f * prm.node_cost);           // :149
    const ObjectPlan object = searchObjectSplit(t, nodeSAH, w);    // :150
    const double tObj = since();

    SpatialPlan spatial; spatial.sah = FLT_MAX; spatial.axis = 0; spatial.pos = 0.0f;
    if (t.level < prm.max_spatial_depth) {                         // :154-162: only when the object split's children overlap enough
        Box overlap = object.lb;
        overlap.clipTo(object.rb);
        if (overlap.area() >= minOverlap) spatial = searchSpatialSplit(t, nodeSAH);
    }

    const double tSpa = since();
    const float minSAH = least(leafSAH, object.sah, spatial.sah);  // :166-168
    if (minSAH == leafSAH && (int64_t)t.n <= (int64_t)prm.max_leaf_size) {
        makeLeaf(node, t.list(2), t.n, w);                 // the object-split search left the stack sorted along z (its last axis)
        return;
    }

    bool split = false;
    if (minSAH == spatial.sah) split = applySpatialSplit(t, spatial, w, left, right);   // :172-174: may leave one side empty ...
    if (!split) applyObjectSplit(t, object, w, left, right);                               // :175-176: ... then the object split is taken
    if (big) std::fprintf(stderr, "sbvh: node n=%u level %d: object search %.3f, spatial search %.3f, split (%s) %.3f s\n", t.n, t.level, tObj, tSpa - tObj, split ? "spatial" : "object", since() - tSpa);
    w.duplicates += (uint64_t)(left->n + right->n - t.n);                               // :180
    left->slot = &node->child[0]; right->slot = &node->child[1];
    left->level = right->level = t.level + 1;
}

// ---------------------------------------------------------------------------------------------- object split
// One axis of the search: candidates "the first i references of the sorted list go left", i = 1 .. n-1, cost
// nodeSAH + area(left box) * cost(i) + area(right box) * cost(n - i) (SplitBVHBuilder.cpp:213-239).  Areas do not depend on the sign
// of a zero bound, so the right-to-left pass may be cut into slices; the boxes of the WINNER are recomputed in the reference's
// order by searchObjectSplit.
void SbvhBuilder::Impl::sweepAxis(const Task& t, int axis, float nodeSAH, Worker& w, ObjectPlan& best) const
{
    const uint32_t n = t.n;
    const uint32_t* list = t.list(axis);
    if (w.areas.size() < n) w.areas.resize(n);
    float* rightArea = w.areas.data();          // rightArea[i]: area of the box of list[i .. n)
    const int parts = helpersFor(n);
    if (parts <= 1) {
        Box rb = Box::empty();
        for (uint32_t i = n - 1; i > 0; i--) { if (i > 8) __builtin_prefetch(&refs[list[i - 8]]); rb.include(refs[list[i]].box); rightArea[i] = rb.area(); }
        Box lb = Box::empty();
        for (uint32_t i = 1; i < n; i++) {
            if (i + 8 < n) __builtin_prefetch(&refs[list[i + 7]]);
            lb.include(refs[list[i - 1]].box);
            const float sah = nodeSAH + lb.area() * triCost(i) + rightArea[i] * triCost(n - i);
            if (sah < best.sah) { best.sah = sah; best.axis = axis; best.numLeft = i; }
        }
        return;
    }
    // big node: slice boxes first, then every slice sweeps with the union of the slices behind / before it
    std::vector<Box> sliceBox((size_t)parts, Box::empty());
    std::vector<size_t> cut((size_t)parts + 1);
    for (int p = 0; p <= parts; p++) cut[(size_t)p] = (size_t)n * (size_t)p / (size_t)parts;
    forSlices((size_t)parts, parts, [&](int, size_t b, size_t e) { for (size_t p = b; p < e; p++) { Box x = Box::empty(); for (size_t i = cut[p]; i < cut[p + 1]; i++) x.include(refs[list[i]].box); sliceBox[p] = x; } });
    std::vector<ObjectPlan> local((size_t)parts);
    forSlices((size_t)parts, parts, [&](int, size_t b, size_t e) {
        for (size_t p = b; p < e; p++) {
            Box rb = Box::empty();
            for (size_t q = (size_t)parts - 1; q > p; q--) rb.include(sliceBox[q]);
            for (size_t i = cut[p + 1]; i-- > cut[p];) { rb.include(refs[list[i]].box); rightArea[i] = rb.area(); }
        }
    });
    forSlices((size_t)parts, parts, [&](int, size_t b, size_t e) {
        for (size_t p = b; p < e; p++) {
            ObjectPlan mine; mine.sah = FLT_MAX; mine.axis = axis; mine.numLeft = 0;
            Box lb = Box::empty();
            for (size_t q = 0; q < p; q++) lb.include(sliceBox[q]);
            for (size_t i = std::max<size_t>(cut[p], 1); i <= cut[p + 1] && i < n; i++) {
                // candidate i of this slice: left box = list[0 .. i)
                if (i - 1 >= cut[p]) lb.include(refs[list[i - 1]].box);
                const float sah = nodeSAH + lb.area() * triCost((uint32_t)i) + rightArea[i] * triCost(n - (uint32_t)i);
                if (sah < mine.sah) { mine.sah = sah; mine.numLeft = (uint32_t)i; }
            }
            local[p] = mine;
        }
    });
    for (int p = 0; p < parts; p++) if (local[(size_t)p].sah < best.sah) { best.sah = local[(size_t)p].sah; best.axis = axis; best.numLeft = local[(size_t)p].numLeft; }
}

ObjectPlan SbvhBuilder::Impl::searchObjectSplit(const Task& t, float nodeSAH, Worker& w) const
{
    ObjectPlan best; best.sah = FLT_MAX; best.axis = 0; best.numLeft = 0; best.lb = Box::empty(); best.rb = Box::empty();
    for (int axis = 0; axis < 3; axis++) sweepAxis(t, axis, nodeSAH, w, best);
    if (best.numLeft) {   // the winner's boxes, united in the order the reference unites them (left: ascending, right: descending)
        const uint32_t* list = t.list(best.axis);
        for (uint32_t i = 0; i < best.numLeft; i++) best.lb.include(refs[list[i]].box);
        for (uint32_t i = t.n; i-- > best.numLeft;) best.rb.include(refs[list[i]].box);
    }
    return best;
}

// Hands the sorted lists down: every reference of the node is marked LEFT or RIGHT; each list (except skipAxis, which is already in
// place) is split stably inside its own slice -- first the numLeft LEFT ones, then the others, both still in sorted order.
void SbvhBuilder::Impl::partitionInPlace(const Task& t, int skipAxis, uint32_t numLeft, Worker& w)
{
    const uint32_t n = t.n;
    const int parts = helpersFor(n);
    if (w.moved.size() < n) w.moved.resize(n);
    uint32_t* tmp = w.moved.data();
    for (int a = 0; a < 3; a++) {
        if (a == skipAxis) continue;
        uint32_t* list = t.list(a);
        if (parts <= 1) {
            uint32_t nl = 0, nr = 0;
            for (uint32_t i = 0; i < n; i++) { const uint32_t id = list[i]; if (refs[id].mark == MARK_LEFT) list[nl++] = id; else tmp[nr++] = id; }
            std::memcpy(list + nl, tmp, (size_t)nr * 4);
            continue;
        }
        std::vector<size_t> nl((size_t)parts + 1, 0);
        forSlices(n, parts, [&](int p, size_t b, size_t e) { size_t c = 0; for (size_t i = b; i < e; i++) c += refs[list[i]].mark == MARK_LEFT; nl[(size_t)p + 1] = c; });
        for (int p = 0; p < parts; p++) nl[(size_t)p + 1] += nl[(size_t)p];
        forSlices(n, parts, [&](int p, size_t b, size_t e) {
            size_t ol = nl[(size_t)p], orr = (size_t)numLeft + (b - nl[(size_t)p]);
            for (size_t i = b; i < e; i++) { const uint32_t id = list[i]; if (refs[id].mark == MARK_LEFT) tmp[ol++] = id; else tmp[orr++] = id; }
        });
        forSlices(n, parts, [&](int, size_t b, size_t e) { std::memcpy(list + b, tmp + b, (e - b) * 4); });
    }
}

void SbvhBuilder::Impl::applyObjectSplit(Task& t, const ObjectPlan& plan, Worker& w, Task*& left, Task*& right)
{
    // performObjectSplit (SplitBVHBuilder.cpp:244-253): sorted along the winning axis, the first numLeft references go left
    left = new Task(); right = new Task();
    left->n = plan.numLeft; right->n = t.n - plan.numLeft;
    left->bounds = plan.lb; right->bounds = plan.rb;
    left->arrivalAxis = right->arrivalAxis = plan.axis;      // both children arrive sorted along that axis
    left->block = right->block = t.block;
    left->offset = t.offset; right->offset = t.offset + plan.numLeft;
    const uint32_t* along = t.list(plan.axis);
    forSlices(t.n, helpersFor(t.n), [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) refs[along[i]].mark = i < plan.numLeft ? MARK_LEFT : MARK_RIGHT; });
    partitionInPlace(t, plan.axis, plan.numLeft, w);
}

// ---------------------------------------------------------------------------------------------- spatial split
// splitReference (SplitBVHBuilder.cpp:445-486): the two parts of a reference cut by the plane x[axis] == pos -- boxes of the triangle's
// vertices on either side plus the edge / plane intersections, each clipped to the reference's current box.
void SbvhBuilder::Impl::chopReference(Box& l, Box& r, const RefRec& ref, int axis, float pos) const
{
    l.clear(); r.clear();
    const int32_t* corner = idx + 3 * (size_t)ref.tri;
    const float* b = verts + 3 * (size_t)corner[2];
    for (int k = 0; k < 3; k++) {            // edges (v2,v0), (v0,v1), (v1,v2)
        const float* a = b;
        b = verts + 3 * (size_t)corner[k];
        const float ap = a[axis], bp = b[axis];
        if (ap <= pos) l.addPoint(a);
        if (ap >= pos) r.addPoint(a);
        if ((ap < pos && bp > pos) || (ap > pos && bp < pos)) {
            const float s = clampFloat((pos - ap) / (bp - ap), 0.0f, 1.0f);
            const float cutPoint[3] = { a[0] * (1.0f - s) + b[0] * s, a[1] * (1.0f - s) + b[1] * s, a[2] * (1.0f - s) + b[2] * s };   // lerp, Sort.h:49
            l.addPoint(cutPoint); r.addPoint(cutPoint);
        }
    }
    l.hi[axis] = pos; r.lo[axis] = pos;
    l.clipTo(ref.box); r.clipTo(ref.box);
}

// findSpatialSplit (SplitBVHBuilder.cpp:265-347): every reference is chopped along the 31 bin planes of each axis; a bin collects the
// boxes of the pieces that fall into it and counts the references that start / end there.  Only areas and counts leave this function,
// so the references may be visited in any order and on several threads.
SpatialPlan SbvhBuilder::Impl::searchSpatialSplit(const Task& t, float nodeSAH) const
{
    struct Bins { Box box[3][kBins]; uint32_t enter[3][kBins], exit[3][kBins], touched[3][kBins]; };
    float origin[3], binSize[3], invBin[3];
    for (int a = 0; a < 3; a++) {
        origin[a] = t.bounds.lo[a];
        binSize[a] = (t.bounds.hi[a] - origin[a]) * (1.0f / (float)kBins);
        invBin[a] = 1.0f / binSize[a];
    }
    const int parts = helpersFor(t.n);
    std::vector<Bins> partial((size_t)parts);
    const uint32_t* list = t.list(0);
    forSlices(t.n, parts, [&](int p, size_t b, size_t e) {
        Bins& bins = partial[(size_t)p];
        for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) { bins.box[a][i].clear(); bins.enter[a][i] = 0; bins.exit[a][i] = 0; bins.touched[a][i] = 0; }
        for (size_t k = b; k < e; k++) {
            const RefRec& ref = refs[list[k]];
            for (int a = 0; a < 3; a++) {
                const int first = clampInt(truncToInt((ref.box.lo[a] - origin[a]) * invBin[a]), 0, kBins - 1);
                const int last = clampInt(truncToInt((ref.box.hi[a] - origin[a]) * invBin[a]), first, kBins - 1);
                RefRec rest = ref;
                for (int i = first; i < last; i++) {
                    Box piece, beyond;
                    chopReference(piece, beyond, rest, a, origin[a] + binSize[a] * (float)(i + 1));
                    bins.box[a][i].include(piece); bins.touched[a][i]++;
                    rest.box = beyond;
                }
                bins.box[a][last].include(rest.box); bins.touched[a][last]++;
                bins.enter[a][first]++;
                bins.exit[a][last]++;
            }
        }
    });
    Bins& bins = partial[0];
    for (int p = 1; p < parts; p++)
        for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) {
            const Bins& o = partial[(size_t)p];
            if (o.touched[a][i]) bins.box[a][i].include(o.box[a][i]);   // an untouched slice bin is not a box (uniting it would make the bin huge)
            bins.enter[a][i] += o.enter[a][i]; bins.exit[a][i] += o.exit[a][i];
        }
    // the sweep over the 31 planes of each axis; EMPTY bins are united like any other (the huge-box rule) -- :318-345
    SpatialPlan best; best.sah = FLT_MAX; best.axis = 0; best.pos = 0.0f;
    for (int a = 0; a < 3; a++) {
        float rightArea[kBins];
        Box rb = Box::empty();
        for (int i = kBins - 1; i > 0; i--) { rb.include(bins.box[a][i]); rightArea[i] = rb.area(); }
        Box lb = Box::empty();
        uint32_t leftNum = 0, rightNum = t.n;
        for (int i = 1; i < kBins; i++) {
            lb.include(bins.box[a][i - 1]);
            leftNum += bins.enter[a][i - 1];
            rightNum -= bins.exit[a][i - 1];
            const float sah = nodeSAH + lb.area() * triCost(leftNum) + rightArea[i] * triCost(rightNum);
            if (sah < best.sah) { best.sah = sah; best.axis = a; best.pos = origin[a] + binSize[a] * (float)i; }
        }
    }
    return best;
}

// performSpatialSplit (SplitBVHBuilder.cpp:349-443).  The reference partitions the node's slice of its stack in place; which references
// it looks at in which order decides the boxes (zero signs), the fate of every straddling reference (the costs compare the boxes and
// counts accumulated SO FAR) and the arrival order of the children.  That order, restated without the stack:
//   * the slice is sorted along z (the object-split search sorted it last).  It is examined like a deque: from the low end -- except
//     that right after a reference lying entirely right of the plane the next one is taken from the HIGH end (the in-place swap pulled
//     it forward);
//   * a reference entirely on the left joins the END of the left block, one entirely on the right the FRONT of the right block;
//   * straddlers wait in a queue in examination order; whenever a left-only reference is met while the queue is not empty, the head of
//     the queue moves to its tail (the swap that makes room for the left block rotates them);
//   * then the queue is emptied from the head: a straddler kept whole on the left joins the end of the left block; kept whole on the
//     right it joins the front of the right block AND the tail of the queue takes over the head position; if it is duplicated, its left
//     part joins the end of the left block and its right part the END of the right block.
// Returns false (nothing changed) when one side ends up empty: the caller then takes the object split, as the reference does.
bool SbvhBuilder::Impl::applySpatialSplit(Task& t, const SpatialPlan& plan, Worker& w, Task*& left, Task*& right)
{
    const int axis = plan.axis; const float pos = plan.pos;
    const uint32_t* sorted = t.list(2);
    std::vector<uint32_t>& leftBlock = w.leftBlock; std::vector<uint32_t>& rightFront = w.rightFront /* stored back to front */; std::vector<uint32_t>& rightTail = w.rightTail;
    std::vector<uint32_t>& waiting = w.waiting;   // the straddler queue: waiting[head ..], grows at the tail only
    leftBlock.clear(); rightFront.clear(); rightTail.clear(); waiting.clear();
    size_t head = 0;
    Box lb = Box::empty(), rb = Box::empty();
    size_t low = 0, high = t.n;
    bool fromHigh = false;
    constexpr size_t kAhead = 12;                          // the records are visited in sorted order, i.e. all over the pool: fetch ahead at both ends
    while (low < high) {
        if (low + kAhead < high) { __builtin_prefetch(&refs[sorted[low + kAhead]]); __builtin_prefetch(&refs[sorted[high - 1 - kAhead]]); }
        const uint32_t id = fromHigh ? sorted[--high] : sorted[low++];
        // a reference pulled forward from the high end sits where the cursor is: the examination simply continues with it
        fromHigh = false;
        const Box& b = refs[id].box;
        if (b.hi[axis] <= pos) {
            lb.include(b); leftBlock.push_back(id);
            if (head < waiting.size()) { const uint32_t first = waiting[head++]; waiting.push_back(first); }
        } else if (b.lo[axis] >= pos) {
            rb.include(b); rightFront.push_back(id);
            fromHigh = true;
        } else waiting.push_back(id);
    }
    std::vector<std::pair<uint32_t, Box>>& madeLeft = w.madeLeft; std::vector<std::pair<uint32_t, Box>>& madeRight = w.madeRight;   // position in its block / box of every new reference (ids are assigned only if the split stands)
    std::vector<uint32_t>& replaced = w.replaced;
    madeLeft.clear(); madeRight.clear(); replaced.clear();
    constexpr uint32_t kNew = 0xFFFFFFFFu;
    while (head < waiting.size()) {
        const uint32_t id = waiting[head];
        const RefRec& ref = refs[id];
        Box lpart, rpart;
        chopReference(lpart, rpart, ref, axis, pos);
        Box lWhole = lb, rWhole = rb, lCut = lb, rCut = rb;
        lWhole.include(ref.box); rWhole.include(ref.box); lCut.include(lpart); rCut.include(rpart);
        const uint32_t nl = (uint32_t)leftBlock.size(), nr = (uint32_t)(rightFront.size() + rightTail.size());
        const float costL = triCost(nl), costR = triCost(nr), costL1 = triCost(nl + 1), costR1 = triCost(nr + 1);
        const float keepLeft = lWhole.area() * costL1 + rb.area() * costR;          // :407-409
        const float keepRight = lb.area() * costL + rWhole.area() * costR1;
        const float cutIt = lCut.area() * costL1 + rCut.area() * costR1;
        const float cheapest = least(keepLeft, keepRight, cutIt);
        if (cheapest == keepLeft) { lb = lWhole; leftBlock.push_back(id); head++; }
        else if (cheapest == keepRight) {
            rb = rWhole; rightFront.push_back(id); head++;
            if (head < waiting.size()) { waiting[--head] = waiting.back(); waiting.pop_back(); }   // the tail takes over the head position
        } else {
            lb = lCut; rb = rCut;
            madeLeft.push_back({ (uint32_t)leftBlock.size(), lpart }); leftBlock.push_back(kNew);
            madeRight.push_back({ (uint32_t)rightTail.size(), rpart }); rightTail.push_back(kNew);
            replaced.push_back(id); head++;
        }
    }
    if (leftBlock.empty() || (rightFront.empty() && rightTail.empty())) return false;

    // the split stands: new references get their ids, the children their lists
    std::vector<KeyedRef>& newLeft = w.newLeft; std::vector<KeyedRef>& newRight = w.newRight;
    newLeft.clear(); newRight.clear();
    for (size_t k = 0; k < replaced.size(); k++) {
        const int32_t tri = refs[replaced[k]].tri;
        const uint32_t a = newRef(w, madeLeft[k].second, tri), b = newRef(w, madeRight[k].second, tri);
        leftBlock[madeLeft[k].first] = a; rightTail[madeRight[k].first] = b;
        newLeft.push_back({ 0, a }); newRight.push_back({ 0, b });
        refs[replaced[k]].mark = MARK_REPLACED;
    }
    left = new Task(); right = new Task();
    left->n = (uint32_t)leftBlock.size(); right->n = (uint32_t)(rightFront.size() + rightTail.size());
    left->bounds = lb; right->bounds = rb;
    forSlices(leftBlock.size(), helpersFor(t.n), [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) { RefRec& r = refs[leftBlock[i]]; if (r.mark != MARK_REPLACED) r.mark = MARK_LEFT; } });
    forSlices(rightFront.size(), helpersFor(t.n), [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) refs[rightFront[i]].mark = MARK_RIGHT; });
    if (replaced.empty()) {                                // nothing was cut: the node's own slices are partitioned in place
        left->block = right->block = t.block;
        left->offset = t.offset; right->offset = t.offset + left->n;
        partitionInPlace(t, -1, left->n, w);
    } else {
        // the node grew: a block of its own; per axis the surviving references (still sorted) are merged with the new ones, which are
        // sorted among themselves first
        std::shared_ptr<ListBlock> grown = std::make_shared<ListBlock>((size_t)left->n + right->n);
        left->block = right->block = grown; left->offset = 0; right->offset = left->n;
        const int sortThreads = helpersFor(t.n);
        for (int a = 0; a < 3; a++) {
            const uint32_t* src = t.list(a);
            for (int side = 0; side < 2; side++) {
                std::vector<KeyedRef>& fresh = side ? newRight : newLeft;
                for (KeyedRef& k : fresh) k.key = sortKey(refs[k.ref], a);
                sortKeyed(fresh, sortThreads, std::min<size_t>(fanOutFrom, (size_t)1 << 16));
                const uint32_t want = side ? MARK_RIGHT : MARK_LEFT;
                uint32_t* dst = (side ? right : left)->list(a);
                size_t j = 0, o = 0;
                for (uint32_t i = 0; i < t.n; i++) {
                    const uint32_t id = src[i];
                    if (refs[id].mark != want) continue;
                    const uint64_t key = sortKey(refs[id], a);
                    while (j < fresh.size() && fresh[j].key < key) dst[o++] = fresh[j++].ref;
                    dst[o++] = id;
                }
                while (j < fresh.size()) dst[o++] = fresh[j++].ref;
            }
        }
    }
    // arrival order: only a child that turns into a leaf without sorting ever looks at it
    left->arrivalAxis = right->arrivalAxis = -1;
    const bool leftNeeds = (int64_t)left->n <= (int64_t)prm.min_leaf_size || t.level + 1 >= prm.max_depth;
    const bool rightNeeds = (int64_t)right->n <= (int64_t)prm.min_leaf_size || t.level + 1 >= prm.max_depth;
    if (leftNeeds) left->arrival = leftBlock;
    if (rightNeeds) {
        right->arrival.assign(rightFront.rbegin(), rightFront.rend());
        right->arrival.insert(right->arrival.end(), rightTail.begin(), rightTail.end());
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- results
float SbvhBuilder::sah() const
{
    // BVHNode::computeSubtreeProbabilities (Source/Nvidia-SBVH/BVHNode.cpp:65-79): pre-order, child 0 first, one running float sum
    float total = 0.0f;
    if (!m->root) return total;
    struct Item { const Node* node; float prob; };
    std::vector<Item> st; st.push_back({ m->root, 1.0f });
    while (!st.empty()) {
        const Item it = st.back(); st.pop_back();
        const Node& n = *it.node;
        const bool leaf = !n.child[0];
        const float cost = (leaf ? 0.0f : 2.0f * m->prm.node_cost) + (leaf ? m->triCost(n.leafCount) : 0.0f);
        total += it.prob * cost;
        if (!leaf) {
            const float pa = n.bounds.area();
            st.push_back({ n.child[1], it.prob * n.child[1]->bounds.area() / pa });
            st.push_back({ n.child[0], it.prob * n.child[0]->bounds.area() / pa });
        }
    }
    return total;
}

void SbvhBuilder::flatten(const uint32_t* vertexMaterial, gmupt_bvh_node* outNodes, gmupt_triangle* outTris, int32_t* outRefTri) const
{
    if (!m->root) return;
    // BVHWrapper.cpp:56-95: explicit stack from the root at slot 0; an inner node reserves the next two slots for its children and the
    // left child is taken up first, so its whole subtree is numbered (and its leaves' triangles are emitted) before the right one's
    struct Item { const Node* node; uint32_t slot; };
    std::vector<Item> st; st.push_back({ m->root, 0u });
    uint32_t slotsUsed = 0, triCount = 0;
    while (!st.empty()) {
        const Item it = st.back(); st.pop_back();
        const Node& n = *it.node;
        gmupt_bvh_node& o = outNodes[it.slot];
        std::memset(&o, 0, sizeof(o));
        for (int k = 0; k < 3; k++) { o.min[k] = n.bounds.lo[k]; o.max[k] = n.bounds.hi[k]; }
        if (!n.child[0]) {
            o.left = (int32_t)triCount; o.right = (int32_t)(triCount + n.leafCount); o.isLeaf = 1;
            for (uint32_t i = 0; i < n.leafCount; i++) {
                const int32_t tri = n.leafTris[i];
                const int32_t* corner = m->idx + 3 * (size_t)tri;
                if (outTris) {
                    gmupt_triangle& T = outTris[triCount];
                    T.v[0] = corner[0]; T.v[1] = corner[1]; T.v[2] = corner[2];
                    T.materialID = vertexMaterial ? vertexMaterial[corner[0]] : 0u;   // BVHWrapper.cpp:82: the first vertex's material
                }
                if (outRefTri) outRefTri[triCount] = tri;
                triCount++;
            }
        } else {
            slotsUsed += 2;
            o.left = (int32_t)slotsUsed - 1; o.right = (int32_t)slotsUsed; o.isLeaf = 0;
            st.push_back({ n.child[1], slotsUsed });
            st.push_back({ n.child[0], slotsUsed - 1 });
        }
    }
}

} // namespace gmupt
