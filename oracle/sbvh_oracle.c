/*
 * oracle/sbvh_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's host-side accel build: the vendored Nvidia split-BVH builder
 * (Source/Nvidia-SBVH/SplitBVHBuilder.cpp, default Platform / BuildParams) followed by BVHWrapper's flatten
 * (Source/BVHWrapper.cpp:56-95).  The product's builder (gmu-path-tracer_amd/host/sbvh_builder.cpp: a different program -- pre-sorted lists, task pool --
 * that must arrive at the same tree) is checked against this node for node.
 *
 * PARITY UNPINNED against the reference binary: the reference builder cannot be compiled here without writing stand-ins
 * for <windows.h> (Source/Nvidia-SBVH/Timer.cpp:30, needed by SplitBVHBuilder's progress timer) and <DirectXMath.h>
 * (Include/BVHWrapper.hpp:3), and the reference ships no tree dumps.  What this file pins is that two independent
 * restatements of the same rules agree.
 *
 * All citations are relative to /root/reference/.
 *
 * This file follows the structure of the vendored builder function by function (that is its purpose: a line-cited restatement to
 * check the product's own builder against), so it carries that code's notice:
 *
 *   Copyright (c) 2009-2011, NVIDIA Corporation.  All rights reserved.
 *
 *   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following
 *   conditions are met:
 *     * Redistributions of source code must retain the above copyright notice, this list of conditions and the following disclaimer.
 *     * Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the following
 *       disclaimer in the documentation and/or other materials provided with the distribution.
 *     * Neither the name of NVIDIA Corporation nor the names of its contributors may be used to endorse or promote products derived
 *       from this software without specific prior written permission.
 *
 *   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT
 *   NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT SHALL
 *   THE COPYRIGHT HOLDER BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT
 *   NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER
 *   CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING
 *   IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
 */
#include <float.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float mn[3], mx[3]; } box_t;
typedef struct { int tri; box_t b; } ref_t;
typedef struct { int numRef; box_t b; } spec_t;
typedef struct { box_t b; int left, right; int lo, hi; } bnode_t; /* left < 0: leaf [lo, hi) of the triangle index list */

typedef struct {
    const float* verts; const int32_t* idx; int numTris;
    ref_t* stack; int stackSize, stackCap;
    box_t* rightBounds;
    bnode_t* nodes; int numNodes, nodeCap;
    int* triIdx; int numTriIdx, triCap;
    float minOverlap;
    int sortDim;
} builder_t;

enum { MAX_DEPTH = 64, MAX_SPATIAL_DEPTH = 48, NUM_BINS = 32 }; /* Include/Nvidia-SBVH/SplitBVHBuilder.h:36-41 */
static const float SPLIT_ALPHA = 1.0e-5f;                        /* Include/Nvidia-SBVH/BVH.h:77 */

/* Include/Nvidia-SBVH/BVHNode.h:41-64 */
static box_t box_empty(void) { box_t b; for (int i = 0; i < 3; i++) { b.mn[i] = FLT_MAX; b.mx[i] = -FLT_MAX; } return b; }
static void box_grow_pt(box_t* b, const float* p) { for (int i = 0; i < 3; i++) { if (!(b->mn[i] < p[i])) b->mn[i] = p[i]; if (!(b->mx[i] > p[i])) b->mx[i] = p[i]; } }
static void box_grow(box_t* b, const box_t* o) { box_t c = *o; box_grow_pt(b, c.mn); box_grow_pt(b, c.mx); } /* grow(min), grow(max): an EMPTY operand blows the box up */
static void box_intersect(box_t* b, const box_t* o) { for (int i = 0; i < 3; i++) { if (!(b->mn[i] > o->mn[i])) b->mn[i] = o->mn[i]; if (!(b->mx[i] < o->mx[i])) b->mx[i] = o->mx[i]; } }
static int box_valid(const box_t* b) { return b->mn[0] <= b->mx[0] && b->mn[1] <= b->mx[1] && b->mn[2] <= b->mx[2]; }
static float box_area(const box_t* b) { if (!box_valid(b)) return 0.0f; float x = b->mx[0] - b->mn[0], y = b->mx[1] - b->mn[1], z = b->mx[2] - b->mn[2]; return (x * y + y * z + z * x) * 2.0f; }

static float min1f(float a, float b) { return (a > b) ? b : a; } /* Include/Nvidia-SBVH/linear_math.h:44 */
static int to_int(float f) { return (f != f || f >= 2147483648.0f || f < -2147483648.0f) ? INT_MIN : (int)f; } /* x86 cvttss2si, Vec3i(Vec3f) linear_math.h:90 */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

static void* grow_array(void* p, int* cap, int need, size_t elem)
{
    if (need <= *cap) return p;
    int c = *cap ? *cap : 64;
    while (c < need) c *= 2;
    *cap = c;
    return realloc(p, (size_t)c * elem);
}
static void push_ref(builder_t* B, const ref_t* r) { B->stack = (ref_t*)grow_array(B->stack, &B->stackCap, B->stackSize + 1, sizeof(ref_t)); B->stack[B->stackSize++] = *r; }
static int new_node(builder_t* B) { B->nodes = (bnode_t*)grow_array(B->nodes, &B->nodeCap, B->numNodes + 1, sizeof(bnode_t)); return B->numNodes++; }

/* SplitBVHBuilder.cpp:103-112: centroid*2 along the sort dimension, ties by triangle index (a strict total order inside a node) */
static builder_t* g_sortCtx;
static int cmp_refs(const void* pa, const void* pb)
{
    const ref_t* a = (const ref_t*)pa; const ref_t* b = (const ref_t*)pb; const int d = g_sortCtx->sortDim;
    const float ca = a->b.mn[d] + a->b.mx[d], cb = b->b.mn[d] + b->b.mx[d];
    return (ca < cb) ? -1 : (ca > cb) ? 1 : (a->tri < b->tri) ? -1 : (a->tri > b->tri) ? 1 : 0;
}
static void sort_tail(builder_t* B, int numRef, int dim) { B->sortDim = dim; g_sortCtx = B; qsort(B->stack + (B->stackSize - numRef), (size_t)numRef, sizeof(ref_t), cmp_refs); }

/* SplitBVHBuilder.cpp:452-490 */
static void split_reference(const builder_t* B, ref_t* left, ref_t* right, const ref_t* ref, int dim, float pos)
{
    left->tri = right->tri = ref->tri;
    left->b = box_empty(); right->b = box_empty();
    const int32_t* inds = B->idx + 3 * (size_t)ref->tri;
    const float* v1 = B->verts + 3 * (size_t)inds[2];
    for (int i = 0; i < 3; i++) {
        const float* v0 = v1;
        v1 = B->verts + 3 * (size_t)inds[i];
        const float v0p = v0[dim], v1p = v1[dim];
        if (v0p <= pos) box_grow_pt(&left->b, v0);
        if (v0p >= pos) box_grow_pt(&right->b, v0);
        if ((v0p < pos && v1p > pos) || (v0p > pos && v1p < pos)) {
            float t = (pos - v0p) / (v1p - v0p);
            t = t < 0.0f ? 0.0f : t > 1.0f ? 1.0f : t;
            float p[3];
            for (int k = 0; k < 3; k++) p[k] = v0[k] * (1.0f - t) + v1[k] * t; /* lerp, Include/Nvidia-SBVH/Sort.h:49 */
            box_grow_pt(&left->b, p); box_grow_pt(&right->b, p);
        }
    }
    left->b.mx[dim] = pos; right->b.mn[dim] = pos;
    box_intersect(&left->b, &ref->b); box_intersect(&right->b, &ref->b);
}

typedef struct { float sah; int dim, numLeft; box_t lb, rb; } objsplit_t;
typedef struct { float sah; int dim; float pos; } spasplit_t;

/* SplitBVHBuilder.cpp:195-241 */
static objsplit_t find_object_split(builder_t* B, const spec_t* spec, float nodeSAH)
{
    objsplit_t best; best.sah = FLT_MAX; best.dim = 0; best.numLeft = 0; best.lb = box_empty(); best.rb = box_empty();
    const int n = spec->numRef;
    for (int dim = 0; dim < 3; dim++) {
        sort_tail(B, n, dim);
        const ref_t* refs = B->stack + (B->stackSize - n);
        box_t rb = box_empty();
        for (int i = n - 1; i > 0; i--) { box_grow(&rb, &refs[i].b); B->rightBounds[i - 1] = rb; }
        box_t lb = box_empty();
        for (int i = 1; i < n; i++) {
            box_grow(&lb, &refs[i - 1].b);
            const float sah = nodeSAH + box_area(&lb) * (float)i + box_area(&B->rightBounds[i - 1]) * (float)(n - i); /* triangle cost 1 */
            if (sah < best.sah) { best.sah = sah; best.dim = dim; best.numLeft = i; best.lb = lb; best.rb = B->rightBounds[i - 1]; }
        }
    }
    return best;
}

/* SplitBVHBuilder.cpp:267-346 */
static spasplit_t find_spatial_split(builder_t* B, const spec_t* spec, float nodeSAH)
{
    static box_t binBox[3][NUM_BINS]; static int binEnter[3][NUM_BINS], binExit[3][NUM_BINS];
    float origin[3], binSize[3], invBin[3];
    for (int d = 0; d < 3; d++) {
        origin[d] = spec->b.mn[d];
        binSize[d] = (spec->b.mx[d] - origin[d]) * (1.0f / (float)NUM_BINS);
        invBin[d] = 1.0f / binSize[d];
        for (int i = 0; i < NUM_BINS; i++) { binBox[d][i] = box_empty(); binEnter[d][i] = 0; binExit[d][i] = 0; }
    }
    for (int ri = B->stackSize - spec->numRef; ri < B->stackSize; ri++) {
        const ref_t ref = B->stack[ri];
        int first[3], last[3];
        for (int d = 0; d < 3; d++) first[d] = clampi(to_int((ref.b.mn[d] - origin[d]) * invBin[d]), 0, NUM_BINS - 1);
        for (int d = 0; d < 3; d++) last[d] = clampi(to_int((ref.b.mx[d] - origin[d]) * invBin[d]), first[d], NUM_BINS - 1);
        for (int d = 0; d < 3; d++) {
            ref_t cur = ref;
            for (int i = first[d]; i < last[d]; i++) {
                ref_t l, r;
                split_reference(B, &l, &r, &cur, d, origin[d] + binSize[d] * (float)(i + 1));
                box_grow(&binBox[d][i], &l.b);
                cur = r;
            }
            box_grow(&binBox[d][last[d]], &cur.b);
            binEnter[d][first[d]]++;
            binExit[d][last[d]]++;
        }
    }
    spasplit_t best; best.sah = FLT_MAX; best.dim = 0; best.pos = 0.0f;
    for (int d = 0; d < 3; d++) {
        box_t rb = box_empty();
        for (int i = NUM_BINS - 1; i > 0; i--) { box_grow(&rb, &binBox[d][i]); B->rightBounds[i - 1] = rb; }
        box_t lb = box_empty(); int leftNum = 0, rightNum = spec->numRef;
        for (int i = 1; i < NUM_BINS; i++) {
            box_grow(&lb, &binBox[d][i - 1]);
            leftNum += binEnter[d][i - 1];
            rightNum -= binExit[d][i - 1];
            const float sah = nodeSAH + box_area(&lb) * (float)leftNum + box_area(&B->rightBounds[i - 1]) * (float)rightNum;
            if (sah < best.sah) { best.sah = sah; best.dim = d; best.pos = origin[d] + binSize[d] * (float)i; }
        }
    }
    return best;
}

static void swap_refs(ref_t* a, ref_t* b) { ref_t t = *a; *a = *b; *b = t; }

/* SplitBVHBuilder.cpp:350-448 */
static void perform_spatial_split(builder_t* B, spec_t* left, spec_t* right, const spec_t* spec, const spasplit_t* split)
{
    const int leftStart = B->stackSize - spec->numRef;
    int leftEnd = leftStart, rightStart = B->stackSize;
    left->b = box_empty(); right->b = box_empty();
    for (int i = leftEnd; i < rightStart; i++) {
        if (B->stack[i].b.mx[split->dim] <= split->pos) { box_grow(&left->b, &B->stack[i].b); swap_refs(&B->stack[i], &B->stack[leftEnd++]); }
        else if (B->stack[i].b.mn[split->dim] >= split->pos) { box_grow(&right->b, &B->stack[i].b); swap_refs(&B->stack[i], &B->stack[--rightStart]); i--; }
    }
    while (leftEnd < rightStart) {
        ref_t lref, rref;
        split_reference(B, &lref, &rref, &B->stack[leftEnd], split->dim, split->pos);
        box_t lub = left->b, rub = right->b, ldb = left->b, rdb = right->b;
        box_grow(&lub, &B->stack[leftEnd].b); box_grow(&rub, &B->stack[leftEnd].b);
        box_grow(&ldb, &lref.b); box_grow(&rdb, &rref.b);
        const float lac = (float)(leftEnd - leftStart), rac = (float)(B->stackSize - rightStart);
        const float lbc = (float)(leftEnd - leftStart + 1), rbc = (float)(B->stackSize - rightStart + 1);
        const float unsplitLeftSAH = box_area(&lub) * lbc + box_area(&right->b) * rac;
        const float unsplitRightSAH = box_area(&left->b) * lac + box_area(&rub) * rbc;
        const float duplicateSAH = box_area(&ldb) * lbc + box_area(&rdb) * rbc;
        const float minSAH = min1f(min1f(unsplitLeftSAH, unsplitRightSAH), duplicateSAH);
        if (minSAH == unsplitLeftSAH) { left->b = lub; leftEnd++; }
        else if (minSAH == unsplitRightSAH) { right->b = rub; swap_refs(&B->stack[leftEnd], &B->stack[--rightStart]); }
        else { left->b = ldb; right->b = rdb; B->stack[leftEnd++] = lref; push_ref(B, &rref); }
    }
    left->numRef = leftEnd - leftStart;
    right->numRef = B->stackSize - rightStart;
}

/* SplitBVHBuilder.cpp:184-191 */
static int create_leaf(builder_t* B, const spec_t* spec)
{
    B->triIdx = (int*)grow_array(B->triIdx, &B->triCap, B->numTriIdx + spec->numRef, sizeof(int));
    for (int i = 0; i < spec->numRef; i++) B->triIdx[B->numTriIdx++] = B->stack[--B->stackSize].tri;
    const int n = new_node(B);
    B->nodes[n].b = spec->b; B->nodes[n].left = B->nodes[n].right = -1;
    B->nodes[n].lo = B->numTriIdx - spec->numRef; B->nodes[n].hi = B->numTriIdx;
    return n;
}

/* SplitBVHBuilder.cpp:123-180 */
static int build_node(builder_t* B, spec_t spec, int level)
{
    if (spec.numRef <= 1 || level >= MAX_DEPTH) return create_leaf(B, &spec); /* Platform minLeafSize 1, Util.h:73 */
    const float area = box_area(&spec.b);
    const float leafSAH = area * (float)spec.numRef;
    const float nodeSAH = area * 2.0f;                                          /* getNodeCost(2) with node cost 1 */
    objsplit_t object = find_object_split(B, &spec, nodeSAH);
    spasplit_t spatial; spatial.sah = FLT_MAX; spatial.dim = 0; spatial.pos = 0.0f;
    if (level < MAX_SPATIAL_DEPTH) {
        box_t overlap = object.lb;
        box_intersect(&overlap, &object.rb);
        if (box_area(&overlap) >= B->minOverlap) spatial = find_spatial_split(B, &spec, nodeSAH);
    }
    const float minSAH = min1f(min1f(leafSAH, object.sah), spatial.sah);
    if (minSAH == leafSAH && spec.numRef <= 0x7FFFFFF) return create_leaf(B, &spec);
    spec_t left, right; left.numRef = right.numRef = 0; left.b = right.b = box_empty();
    if (minSAH == spatial.sah) perform_spatial_split(B, &left, &right, &spec, &spatial);
    if (!left.numRef || !right.numRef) {                                        /* performObjectSplit :245-254 */
        sort_tail(B, spec.numRef, object.dim);
        left.numRef = object.numLeft; left.b = object.lb;
        right.numRef = spec.numRef - object.numLeft; right.b = object.rb;
    }
    const int rightNode = build_node(B, right, level + 1);                      /* :176-177: right first */
    const int leftNode = build_node(B, left, level + 1);
    const int n = new_node(B);
    B->nodes[n].b = spec.b; B->nodes[n].left = leftNode; B->nodes[n].right = rightNode; B->nodes[n].lo = B->nodes[n].hi = 0;
    return n;
}

typedef struct { float min[3]; float pad0; float max[3]; float pad1; int32_t left, right, isLeaf; float pad2; } flat_node_t; /* 48 B */
typedef struct { int32_t v[3]; uint32_t materialID; } flat_tri_t;                                                               /* 16 B */

/*
 * Builds and flattens.  out_nodes / out_tris are malloc'ed (free with orc_sbvh_free).  Returns the node count (0 on failure).
 */
int orc_sbvh_build(const float* verts, int numVerts, const int32_t* idx, int numTris, const uint32_t* vertexMaterial,
                   flat_node_t** out_nodes, flat_tri_t** out_tris, int* out_numRefs)
{
    (void)numVerts;
    builder_t B; memset(&B, 0, sizeof(B));
    B.verts = verts; B.idx = idx; B.numTris = numTris;
    spec_t root; root.numRef = numTris; root.b = box_empty();
    for (int i = 0; i < numTris; i++) {                                         /* SplitBVHBuilder.cpp:63-78 */
        ref_t r; r.tri = i; r.b = box_empty();
        for (int j = 0; j < 3; j++) box_grow_pt(&r.b, verts + 3 * (size_t)idx[3 * (size_t)i + j]);
        push_ref(&B, &r);
        box_grow(&root.b, &r.b);
    }
    B.minOverlap = box_area(&root.b) * SPLIT_ALPHA;
    const int rbN = (numTris > NUM_BINS ? numTris : NUM_BINS) - 1;
    B.rightBounds = (box_t*)malloc(sizeof(box_t) * (size_t)(rbN > 0 ? rbN : 1));
    const int rootNode = build_node(&B, root, 0);

    /* BVHWrapper.cpp:56-95 flatten: explicit stack, children of an inner node in consecutive slots */
    flat_node_t* nodes = (flat_node_t*)calloc((size_t)B.numNodes, sizeof(flat_node_t));
    flat_tri_t* tris = (flat_tri_t*)calloc((size_t)(B.numTriIdx ? B.numTriIdx : 1), sizeof(flat_tri_t));
    int (*st)[2] = (int (*)[2])malloc(sizeof(int[2]) * (size_t)(B.numNodes + 1));
    int sp = 0, nodeIndex = 0, triCount = 0;
    st[sp][0] = rootNode; st[sp][1] = 0; sp++;
    while (sp > 0) {
        sp--;
        const bnode_t* n = &B.nodes[st[sp][0]]; const int cur = st[sp][1];
        flat_node_t* o = &nodes[cur];
        for (int k = 0; k < 3; k++) { o->min[k] = n->b.mn[k]; o->max[k] = n->b.mx[k]; }
        if (n->left < 0) {
            o->left = triCount; o->right = triCount + (n->hi - n->lo); o->isLeaf = 1;
            for (int i = n->lo; i < n->hi; i++) {
                const int32_t* ind = idx + 3 * (size_t)B.triIdx[i];
                tris[triCount].v[0] = ind[0]; tris[triCount].v[1] = ind[1]; tris[triCount].v[2] = ind[2];
                tris[triCount].materialID = vertexMaterial ? vertexMaterial[ind[0]] : 0u; /* BVHWrapper.cpp:82 */
                triCount++;
            }
        } else {
            nodeIndex += 2;
            st[sp][0] = n->right; st[sp][1] = nodeIndex; sp++; o->right = nodeIndex;
            st[sp][0] = n->left; st[sp][1] = nodeIndex - 1; sp++; o->left = nodeIndex - 1;
            o->isLeaf = 0;
        }
    }
    const int numNodes = B.numNodes;
    free(st); free(B.stack); free(B.rightBounds); free(B.nodes); free(B.triIdx);
    *out_nodes = nodes; *out_tris = tris; *out_numRefs = triCount;
    return numNodes;
}

void orc_sbvh_free(void* p) { free(p); }
