// The trace loop by tree levels ("wavefront" pipeline): the alternative to k_render's one-lane-per-pixel
// state machine.  Included by rtx_hip.hip (inside its unnamed namespace, after DevCamera, DevFrame,
// primaryDir, toByte and flushCounters).
//
// k_render keeps a pixel's whole call tree (Scene::traceRay, src/Scene.cpp:83-106) on one lane: a chain
// of closest-hit ray, shadow ray per light, children after parents.  The frame can then be no shorter
// than the longest chain, and the lanes of a wave sit in different phases of different trees.  Here the
// unit of scheduling is a RAY.  A strip of the frame is traced level by level (level l = the traceRay
// calls of remaining depth D - l), three launches per level, every one over dense lists of like work:
//
//   k_wf_trace   one lane per trace request: sceneIntersect.  Miss: the sky colour goes to the parent.
//                Hit: a node record (point, normal, view, material) is appended to the level's node list
//                and one shadow request per light that can contribute (aimLight) to the shadow lists.
//   k_wf_shadow  one lane per shadow request: the shadow ray of calcLightning (:65-72); sets the light's
//                bit in the node's `hidden` mask.  (k_wf_shadow_pool, for scenes whose spheres sit behind a
//                tree: the same, with the lanes of a wave drawing on the list one by one as they finish.)
//   k_wf_shade   one lane per node: adds up the lights in the reference's order (:73-78), forms the
//                node's own colour, and appends the reflect / refract rays (:95-101) to the next level's
//                request lists.  Nodes of the last level take their children's sky colours in place.
//
// and then bottom-up, k_wf_resolve per level: colour = (own + reflect * a2) + refract * a3 exactly as at
// :103-105, delivered into the parent's record; k_wf_pixels turns the strip's sample colours into B,G,R
// bytes.
//
// Regrouping.  Every append is a ballot/mbcnt compaction with one atomic per wave, and every request is
// CLASSED when it is made: rays that pass the box around the mesh go to one set of lists, all others to
// a second set.  A wave of the consuming launch then holds rays of one class only -- 64 lanes that all
// walk the mesh, or 64 lanes none of which does -- instead of a few walkers scattered over every wave.
//
// All per-ray arithmetic is the same device code k_render runs (castBegin / castMesh / castEnd, aimLight,
// pathAddLight, skyColor ...), in the same order per node: frames are byte-identical.

constexpr int kWfLevels = kMaxDepth;
constexpr unsigned kLinkIndex = 0x1fffffffu;  // bits 0..28: index of the parent's record (or of the sample, level 0)
constexpr unsigned kLinkRefr = 1u << 29;      // the request is the parent's refract child (else its reflect child)
constexpr unsigned kHasRefl = 1u << 30, kHasRefr = 1u << 31;  // thin record: which children were traced
constexpr unsigned kWfMaxIndex = kLinkIndex;
constexpr int kWfMaxLights = 32;              // bits of the cast / hidden masks; shadow request = node << light_bits | light

// Every list (trace requests, nodes, shadow requests of a level) is a set of sub-lists, each with its own
// length counter in a cache line of its own: atomics on a single address go one after the other at about
// 10 ns each, and a 4K frame makes half a million appends.  Work is handed out in chunks of 64 entries (one
// wave); the wave that works on chunk c appends to sub-list c % kWfSub of its output lists, so no sub-list
// can get more than its share of the worst case, and entry (k, j) lives at index k * sub_cap + j of the
// list's arrays.  Request and shadow lists come in two classes: sub-lists 0 .. kWfSub-1 hold the rays that
// pass the mesh's box, kWfSub .. 2 kWfSub-1 the others; node lists have one class.  The two classes of one k
// share one region of sub_cap entries: the first fills it from the front, the second from the back.
constexpr unsigned kWfSub = 64, kWfMaxLists = 2 * kWfSub, kWfCtrStride = 32, kWfChunk = 64;
struct WfList {
    unsigned *count;   // n_sub lengths, kWfCtrStride words apart
    unsigned sub_cap;  // entries a sub-list has room for
    unsigned n_sub;    // kWfSub or 2 * kWfSub
};
struct WfCounters {
    unsigned n[3][kWfLevels + 1][kWfMaxLists * kWfCtrStride];  // [0] trace requests, [1] nodes, [2] shadow requests of level l
    unsigned work[4][kWfLevels + 1][kWfSub * kWfCtrStride];    // chunk hand-out of the level's trace / shadow / shade / resolve launch
};

// zeroes the counters of levels 0 .. levels (only the words in use: one per cache line)
__global__ __launch_bounds__(kBlock) void k_wf_clear(WfCounters *ctr, int levels) {
    const unsigned per_level = 3u * kWfMaxLists + 4u * kWfSub;
    for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < per_level * (unsigned)(levels + 1); t += gridDim.x * kBlock) {
        const unsigned l = t / per_level, k = t % per_level;
        if (k < 3u * kWfMaxLists) ctr->n[k / kWfMaxLists][l][(k % kWfMaxLists) * kWfCtrStride] = 0u;
        else ctr->work[(k - 3u * kWfMaxLists) / kWfSub][l][((k - 3u * kWfMaxLists) % kWfSub) * kWfCtrStride] = 0u;
    }
}

// A strip = tiles [first_tile, first_tile + n_tiles) of the frame's 8x8 tiles in row-major tile order; sample
// i of the strip is sample i % spp of pixel (i / spp) % 64 of tile first_tile + i / (64 * spp).
struct WfStrip {
    unsigned first_tile, n_tiles, tiles_x;
};

// Records of one node, in float4 so that every access is one 16-byte load or store.
//   fat  (lives from the level's trace launch to its shade launch, one set of arrays for all levels; an array per
//        field, so that the lanes of a wave -- neighbouring nodes -- read and write whole cache lines, and k_wf_shadow,
//        which wants the first two fields only, fetches nothing else):
//        pt  point.xyz, material index   nrm  N.xyz, cast mask
//        vw  view.xyz                    dif  diffuse.rgb (after water / island override), link to the parent
//        hidden  one word: the lights found hidden (k_wf_shadow ORs into it; a cache line holds 32 nodes' words)
//   thin (lives until the level is resolved; per level four arrays in one block of thin_n records each: what a launch
//        does not need it does not fetch, and a child's colour is one whole 16-byte store):
//        own   own colour.rgb, a2                  kid0  reflect child's colour.rgb (written by the child)
//        meta  a3, link to the parent | kHasRefl | kHasRefr (8 bytes)   kid1  refract child's colour.rgb
struct WfArgs {
    WfList req, nodes, shadows, next_req;  // this level's requests, nodes, shadow requests; the next level's requests
    unsigned *work;                        // this launch's chunk hand-out counters
    const float4 *ray_o, *ray_d;           // this level's requests: orig.xyz + link, dir.xyz
    float4 *next_o, *next_d;               // the next level's
    float4 *pt, *nrm, *vw, *dif;
    unsigned *hidden;
    unsigned *shadow_q;
    float4 *thin;                          // this level's thin records: own at [0, thin_n), kid0 and kid1 after it, then meta
    float4 *parent;                        // the level above's thin records, or the strip's sample colours (level 0)
    unsigned thin_n, parent_n;             // records per array of the two
    int level;
    unsigned light_bits;                   // bits of a shadow request that hold the light's number (as many as the scene's lights need)
};

RTX_DEV unsigned laneId() { return __builtin_amdgcn_mbcnt_hi(0xffffffffu, __builtin_amdgcn_mbcnt_lo(0xffffffffu, 0u)); }
RTX_DEV unsigned lanesBelow(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// the colour of a finished traceRay call goes where its caller reads it
RTX_DEV void wfDeliver(float4 *parent, unsigned parent_n, int level, unsigned link, V3 c) {
    // level 0: the sample's colour; below: the parent's kid0 (reflect) or kid1 (refract) entry
    const size_t at = level == 0 ? 0 : (size_t)parent_n * ((link & kLinkRefr) ? 2u : 1u);
    parent[at + (link & kLinkIndex)] = make_float4(c.x, c.y, c.z, 0.0f);
}
RTX_DEV uint2 *wfThinMeta(float4 *thin, unsigned thin_n) { return reinterpret_cast<uint2 *>(thin + 3 * (size_t)thin_n); }

// does the ray pass the (slightly widened) box around the mesh?  Only decides which list a request goes to.
RTX_DEV bool wfMeshClass(const DevScene &s, V3 orig, V3 dir) {
    return s.n_mesh_tris > 0 && slabMaybe(s.mesh_lo, s.mesh_hi, orig, safeInverse(dir));
}

// A workgroup's view of a list as chunks of kWfChunk entries: every sub-list is cut into chunks of its own
// (the last one partly filled), numbered sub-list after sub-list.
struct WfChunks {
    unsigned first[kWfMaxLists + 1];  // number of the first chunk of each sub-list; [kWfMaxLists] = chunks in all
    unsigned len[kWfMaxLists];
    unsigned n_chunks;
};
// by all threads of the workgroup, once per launch (ends with a barrier)
RTX_DEV void wfChunksLoad(WfChunks &ch, const WfList &list) {
    if (threadIdx.x < 64) {  // wave 0: two sub-lists per lane
        const unsigned k0 = 2 * threadIdx.x, k1 = k0 + 1;
        const unsigned n0 = k0 < list.n_sub ? list.count[k0 * kWfCtrStride] : 0u, n1 = k1 < list.n_sub ? list.count[k1 * kWfCtrStride] : 0u;
        const unsigned c0 = (n0 + kWfChunk - 1) / kWfChunk, c1 = (n1 + kWfChunk - 1) / kWfChunk;
        unsigned incl = c0 + c1;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(incl, off);
            if ((int)threadIdx.x >= off) incl += up;
        }
        ch.len[k0] = n0, ch.len[k1] = n1;
        ch.first[k0] = incl - c0 - c1, ch.first[k1] = incl - c1;
        if (threadIdx.x == 63) ch.first[kWfMaxLists] = incl, ch.n_chunks = incl;
    }
    __syncthreads();
}
// entry e (0 .. kWfChunk - 1) of chunk c: false if the chunk has no such entry
RTX_DEV bool wfChunkEntryAt(const WfChunks &ch, const WfList &list, unsigned c, unsigned e, unsigned &index) {
    unsigned k = 0;
    for (unsigned step = kWfMaxLists / 2; step; step >>= 1)  // largest k with first[k] <= c
        if (ch.first[k + step] <= c) k += step;
    const unsigned j = (c - ch.first[k]) * kWfChunk + e;
    index = k < kWfSub ? k * list.sub_cap + j : (k - kWfSub + 1u) * list.sub_cap - 1u - j;
    return j < ch.len[k];
}
// entry of chunk c that this lane works on: false if the chunk has none for it
RTX_DEV bool wfChunkEntry(const WfChunks &ch, const WfList &list, unsigned c, unsigned &index) { return wfChunkEntryAt(ch, list, c, laneId(), index); }

// Chunk hand-out, per wave: the chunk numbers are dealt into kWfSub interleaved ranges (range r = chunks r,
// r + kWfSub, ...), each with a counter of its own; a wave counts on the range it starts from and moves on
// to the next range that still has chunks when its own runs dry.  `r` is the wave's current range.
// `ahead` holds the count taken for the NEXT call (lane 0; asked for one chunk early so that the atomic's
// round trip runs under the current chunk's work): start with wfTakeFirst.
RTX_DEV unsigned wfTakeFirst(unsigned *work, unsigned r) {
    unsigned j = 0;
    if (laneId() == 0) j = atomicAdd(work + r * kWfCtrStride, 1u);
    return j;
}
RTX_DEV bool wfTakeChunk(unsigned *work, unsigned n_chunks, unsigned &r, unsigned &ahead, unsigned &chunk) {
    const unsigned lane = laneId();
    for (;;) {
        const unsigned j = (unsigned)__builtin_amdgcn_readfirstlane((int)ahead);
        chunk = r + j * kWfSub;
        if (chunk < n_chunks) {
            ahead = wfTakeFirst(work, r);
            return true;
        }
        // this range is used up: every lane looks at one range's counter
        const unsigned taken = __hip_atomic_load(work + lane * kWfCtrStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long live = __ballot((unsigned long long)lane + (unsigned long long)taken * kWfSub < n_chunks);
        live &= ~(1ull << r);
        if (!live) return false;
        const unsigned long long after = r == 63u ? 0ull : live & (~0ull << (r + 1u));
        r = (unsigned)__builtin_ctzll(after ? after : live);
        ahead = wfTakeFirst(work, r);
    }
}

// Appends the wave's entries to the two classes of a list: lanes with `want` and cls = 0 / 1.  One atomic per
// class with entries, issued back to back.  Returns the lane's index in the list's arrays.
RTX_DEV unsigned wfAppendClassed(const WfList &list, unsigned chunk, bool want, bool mesh_class) {
    const unsigned long long m0 = __ballot(want && mesh_class), m1 = __ballot(want && !mesh_class);
    const unsigned k = chunk % kWfSub;
    unsigned b0 = 0u, b1 = 0u;
    if (laneId() == 0) {
        if (m0) b0 = atomicAdd(list.count + k * kWfCtrStride, (unsigned)__popcll(m0));
        if (m1) b1 = atomicAdd(list.count + (kWfSub + k) * kWfCtrStride, (unsigned)__popcll(m1));
    }
    b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)b0), b1 = (unsigned)__builtin_amdgcn_readfirstlane((int)b1);
    return mesh_class ? k * list.sub_cap + b0 + lanesBelow(m0) : (k + 1u) * list.sub_cap - 1u - (b1 + lanesBelow(m1));
}

// The pixel of sample i of the strip; false for the samples of an edge tile that lie outside the frame.
RTX_DEV bool wfSamplePixel(const DevFrame &fr, const WfStrip &strip, unsigned i, int &row, int &col, int &k) {
    const unsigned spp = (unsigned)fr.spp;
    k = (int)(i % spp);
    const unsigned p = i / spp;
    const unsigned tile = strip.first_tile + (p >> 6), within = p & 63u;
    const int px = (int)(tile % strip.tiles_x) * 8 + (int)(within & 7u), pk = (int)(tile / strip.tiles_x) * 8 + (int)(within >> 3);
    col = px, row = fr.row_first + pk * fr.row_step;
    return px < fr.width && pk < fr.rows;
}

// The traversal stacks live in dynamic LDS, sized by the launch for the depth of the scene's trees
// (stack entries x kBlock ints): the less of it, the more workgroups a CU holds.
extern __shared__ int s_wf_stack[];
// workgroups per CU of the general build's walking kernels
#ifndef RTX_WF_TRACE_BLOCKS
#define RTX_WF_TRACE_BLOCKS 5
#endif
#ifndef RTX_WF_SHADOW_BLOCKS
#define RTX_WF_SHADOW_BLOCKS 5
#endif

template <bool STATS, bool WIDE>
__global__ __launch_bounds__(kBlock, STATS ? 2 : WIDE ? RTX_WF_TRACE_BLOCKS : 4) void k_wf_trace(DevScene s, DevCamera cam, DevFrame fr, WfStrip strip, WfArgs a,
                                                                    Counters *counters) {
    __shared__ SmallScene q;
    __shared__ WfChunks ch;
    loadSmallScene(q, s);
    int *stack = s_wf_stack + threadIdx.x;
    const int level = a.level, depth = fr.depth - level;  // remaining depth of this level's traceRay calls
    const unsigned n_samples = strip.n_tiles * 64u * (unsigned)fr.spp;
    unsigned n_chunks = (n_samples + kWfChunk - 1) / kWfChunk;  // level 0: the strip's samples, no list
    if (level != 0) {
        wfChunksLoad(ch, a.req);
        n_chunks = ch.n_chunks;
    }
    __syncthreads();
    LaneCounters<STATS> lc;
    unsigned range = (blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6)) % kWfSub, chunk;
    unsigned ahead = wfTakeFirst(a.work, range);
    while (wfTakeChunk(a.work, n_chunks, range, ahead, chunk)) {
        // (nothing read from memory is carried over from one chunk to the next: without this the compiler lifts the LDS
        // reads of scene constants out of the loop and spills to make room -- k_wf_trace 40 -> 24 spilled registers, the
        // stress scene's frame 88.1 -> 84.8 ms and 117 -> 87 GB of HBM-side traffic)
        asm volatile("" ::: "memory");
        unsigned i = chunk * kWfChunk + laneId();
        bool valid = i < n_samples;
        if (level != 0) valid = wfChunkEntry(ch, a.req, chunk, i);
        V3 orig = v3(0, 0, 0), dir = v3(0, 0, 1);
        unsigned link = i;
        if (level == 0) {
            int row, col, k;
            valid = valid && wfSamplePixel(fr, strip, i, row, col, k);
            if (valid) {
                orig = v3(cam.ox, cam.oy, cam.oz);
                if (WIDE && fr.spp > 1) dir = primaryDir(cam, fr.width, fr.height, row, col, sampleDx(k), sampleDy(k));
                else dir = primaryDir(cam, fr.width, fr.height, row, col);
            }
        } else if (valid) {
            const float4 o = a.ray_o[i], d = a.ray_d[i];
            orig = v3(o.x, o.y, o.z), dir = v3(d.x, d.y, d.z), link = __float_as_uint(o.w);
        }
        const bool casts = valid && depth > 0;  // a call of depth 0 is answered by the sky (src/Scene.cpp:87)
        if constexpr (STATS) {
            lc.v[CNT_PRIMARY] += casts && level == 0 ? 1u : 0u;
            lc.v[CNT_SECONDARY] += casts && level != 0 ? 1u : 0u;
        }
        Hit h;
        float md = 0.0f;
        bool hit = false;
        if (casts) hit = sceneIntersect<STATS, WIDE, true>(s, q, orig, dir, false, -1.0f, h, md, stack, kBlock, lc);
        if (valid && !hit) {
            RTX_COUNT(CNT_SKY);
            wfDeliver(a.parent, a.parent_n, level, link, skyColor(s, dir));
        }
        // ---- a node per hit; a shadow request per light that can contribute, classed by its ray ----
        const bool node_here = valid && hit;
        unsigned cast_mask = 0, mesh_mask = 0;
        if (node_here) {
            PathState st;
            st.point = h.point, st.N = h.N, st.m = h.m, st.view = dir;
            for (int li = 0; li < s.n_lights; ++li) {
                const LightAim aim = aimLight<WIDE>(s, q, st, li);
                if (!aim.counts) continue;
                cast_mask |= 1u << li;
                const V3 off = st.N * s.fprec;  // the shadow ray's origin, as pathAimShadowRay forms it
                if (wfMeshClass(s, __builtin_signbit(aim.ndl) ? st.point - off : st.point + off, aim.dir)) mesh_mask |= 1u << li;
            }
        }
        // the wave's appends: the counts first, then all atomics back to back (one round trip)
        const unsigned long long mn = __ballot(node_here);
        const unsigned k = chunk % kWfSub;
        unsigned n0 = 0, n1 = 0;
        for (int li = 0; li < s.n_lights; ++li) {
            n0 += (unsigned)__popcll(__ballot(((cast_mask & mesh_mask) >> li) & 1u));
            n1 += (unsigned)__popcll(__ballot(((cast_mask & ~mesh_mask) >> li) & 1u));
        }
        unsigned node_at = 0, s0 = 0, s1 = 0;
        if (laneId() == 0) {
            if (mn) node_at = atomicAdd(a.nodes.count + k * kWfCtrStride, (unsigned)__popcll(mn));
            if (n0) s0 = atomicAdd(a.shadows.count + k * kWfCtrStride, n0);
            if (n1) s1 = atomicAdd(a.shadows.count + (kWfSub + k) * kWfCtrStride, n1);
        }
        node_at = (unsigned)__builtin_amdgcn_readfirstlane((int)node_at);
        s0 = k * a.shadows.sub_cap + (unsigned)__builtin_amdgcn_readfirstlane((int)s0);
        s1 = (k + 1u) * a.shadows.sub_cap - 1u - (unsigned)__builtin_amdgcn_readfirstlane((int)s1);  // the second class fills its region backwards
        const unsigned node = k * a.nodes.sub_cap + node_at + lanesBelow(mn);
        if (node_here) {
            a.pt[node] = make_float4(h.point.x, h.point.y, h.point.z, __int_as_float(h.mat_index));
            a.nrm[node] = make_float4(h.N.x, h.N.y, h.N.z, __uint_as_float(cast_mask));
            a.vw[node] = make_float4(dir.x, dir.y, dir.z, 0.0f);
            a.dif[node] = make_float4(h.m.dr, h.m.dg, h.m.db, __uint_as_float(link));
            a.hidden[node] = 0u;
        }
        // in the lists the requests stand light by light, so that the lanes of a wave of k_wf_shadow aim at one
        // light from neighbouring points
        if (n0 | n1) {
            for (int li = 0; li < s.n_lights; ++li) {
                const bool c0 = ((cast_mask & mesh_mask) >> li) & 1u, c1 = ((cast_mask & ~mesh_mask) >> li) & 1u;
                const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
                if (c0) a.shadow_q[s0 + lanesBelow(m0)] = (node << a.light_bits) | (unsigned)li;
                if (c1) a.shadow_q[s1 - lanesBelow(m1)] = (node << a.light_bits) | (unsigned)li;
                s0 += (unsigned)__popcll(m0), s1 -= (unsigned)__popcll(m1);
            }
        }
    }
    flushCounters<STATS>(lc, counters);
}

template <bool STATS, bool WIDE>
__global__ __launch_bounds__(kBlock, STATS ? 2 : WIDE ? RTX_WF_SHADOW_BLOCKS : 4) void k_wf_shadow(DevScene s, WfArgs a, Counters *counters) {
    __shared__ SmallScene q;
    __shared__ WfChunks ch;
    loadSmallScene(q, s);
    int *stack = s_wf_stack + threadIdx.x;
    wfChunksLoad(ch, a.shadows);
    const unsigned n_chunks = ch.n_chunks;
    LaneCounters<STATS> lc;
    unsigned range = (blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6)) % kWfSub, chunk;
    unsigned ahead = wfTakeFirst(a.work, range);
    while (wfTakeChunk(a.work, n_chunks, range, ahead, chunk)) {
        asm volatile("" ::: "memory");  // (as in k_wf_trace)
        unsigned i;
        const bool valid = wfChunkEntry(ch, a.shadows, chunk, i);
        if (!valid) continue;
        const unsigned e = a.shadow_q[i];
        const unsigned node = e >> a.light_bits, li = e & ((1u << a.light_bits) - 1u);
        const float4 f0 = a.pt[node], f1 = a.nrm[node];
        PathState st;
        st.point = v3(f0.x, f0.y, f0.z), st.N = v3(f1.x, f1.y, f1.z);
        // the same three values aimLight produced when the request was made (src/Scene.cpp:65)
        const float4 L = lightAt<WIDE>(s, q, (int)li);
        LightAim aim;
        aim.pos = v3(L.x, L.y, L.z), aim.intensity = L.w;
        const V3 to_light = aim.pos - st.point;
        aim.dist = sqrtf(dot(to_light, to_light));
        aim.dir = to_light * (1.0f / aim.dist);
        aim.ndl = dot(aim.dir, st.N);
        pathAimShadowRay(s, st, aim);
        RTX_COUNT(CNT_SHADOW);
        Hit h;
        float md = 0.0f;
        const bool hit = sceneIntersect<STATS, WIDE, true>(s, q, st.orig, st.dir, true, st.t_exit, h, md, stack, kBlock, lc);
        if (pathShadowed(st, hit, md)) atomicOr(a.hidden + node, 1u << li);
    }
    flushCounters<STATS>(lc, counters);
}

// k_wf_shadow for scenes whose spheres sit behind a tree: the lanes of a wave do not work through a chunk of the
// list in step -- walks differ in length, and the wave would wait for its longest -- but draw on it one by one.  A
// lane is idle, walking (the sphere tree: by far the longest part of a shadow ray here) or has walked; the wave
// walks until fewer than kWfPoolLow lanes do, then the lanes that have walked finish their rays (the other
// categories, the mesh, the verdict), idle lanes take the next entries of the list and start their walks, and the
// walk goes on with a full wave.  The part around the walks so runs for more than half a wave at a time, the walk
// itself for between 28 lanes and all 64.  Same rays, same tests per ray.  Measured on 10 000 spheres (4096 x 4096,
// 4 samples): 43 instead of 31 lanes per vector instruction, a quarter fewer node trips, and the part around the
// walks -- some thousand instructions a ray, now run 1.8 times as often -- two thirds dearer: 45.7 -> 41.8 ms per
// frame for the shadow launches (threshold 16 / 28 / 40 / 52 lanes: frame 89.4 / 87.9 / 88.9 / 93.0 ms).
#ifndef RTX_WF_POOL_BLOCKS
#define RTX_WF_POOL_BLOCKS 5
#endif
#ifndef RTX_WF_POOL_LOW
#define RTX_WF_POOL_LOW 28
#endif
constexpr unsigned kWfPoolLow = RTX_WF_POOL_LOW;
template <bool STATS>
__global__ __launch_bounds__(kBlock, STATS ? 2 : RTX_WF_POOL_BLOCKS) void k_wf_shadow_pool(DevScene s, WfArgs a, Counters *counters) {
    __shared__ SmallScene q;
    __shared__ WfChunks ch;
    loadSmallScene(q, s);
    int *stack = s_wf_stack + threadIdx.x;
    wfChunksLoad(ch, a.shadows);
    const unsigned n_chunks = ch.n_chunks;
    LaneCounters<STATS> lc;
    unsigned range = (blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6)) % kWfSub, chunk = 0;
    unsigned ahead = wfTakeFirst(a.work, range);
    unsigned dealt = kWfChunk;  // entries of `chunk` handed to lanes so far (the same in all lanes)
    bool more = true;           // the list may have chunks left
    enum { kIdle, kWalking, kWalked };
    int state = kIdle;
    PathState st;               // the shadow ray of this lane: orig, dir, t_exit, lightDist
    unsigned node = 0, li = 0;
    bool gated = false;         // the ray passes the box around the spheres
    SphereWalker w;
    w.cur = w.pend = kWalkDone, w.sp = 0;
    float minDist = kFltMax;
    int index = -1;
    for (;;) {
        asm volatile("" ::: "memory");  // (as in k_wf_trace)
        // ---- the rays that have walked: the rest of sceneIntersect, the verdict ----
        if (state == kWalked) {
            const V3 inv = safeInverse(st.dir);
            CastState cs;
            cs.minDist = kFltMax, cs.kind = 0, cs.index = -1;
            bool mesh = false;
            if (gated) mesh = castLoose<STATS, true>(s, q, st.orig, st.dir, inv, st.t_exit, minDist, index >= 0 ? 1 : 0, index, cs, lc);
            if (mesh) castMesh<STATS, true>(s, st.orig, st.dir, st.t_exit, cs, stack, kBlock, lc);
            Hit h;
            float md = 0.0f;
            const bool hit = castEnd<STATS, true>(s, q, st.orig, st.dir, true, st.t_exit, cs, h, md, lc);
            if (pathShadowed(st, hit, md)) atomicOr(a.hidden + node, 1u << li);
            state = kIdle;
        }
        // ---- idle lanes take the next entries of the list ----
        unsigned entry = 0;
        bool got = false;
        for (;;) {
            const unsigned long long idle = __ballot(state == kIdle && !got);
            if (!idle) break;
            if (dealt >= kWfChunk) {
                if (!more || !wfTakeChunk(a.work, n_chunks, range, ahead, chunk)) {
                    more = false;
                    break;
                }
                dealt = 0;
            }
            const unsigned e = dealt + lanesBelow(idle);
            if (state == kIdle && !got && e < kWfChunk) got = wfChunkEntryAt(ch, a.shadows, chunk, e, entry);
            dealt = min(kWfChunk, dealt + (unsigned)__popcll(idle));
        }
        // ---- their rays, up to the start of the walk ----
        if (got) {
            const unsigned req = a.shadow_q[entry];
            node = req >> a.light_bits, li = req & ((1u << a.light_bits) - 1u);
            const float4 f0 = a.pt[node], f1 = a.nrm[node];
            st.point = v3(f0.x, f0.y, f0.z), st.N = v3(f1.x, f1.y, f1.z);
            // the same three values aimLight produced when the request was made (src/Scene.cpp:65)
            const float4 L = lightAt<true>(s, q, (int)li);
            LightAim aim;
            aim.pos = v3(L.x, L.y, L.z), aim.intensity = L.w;
            const V3 to_light = aim.pos - st.point;
            aim.dist = sqrtf(dot(to_light, to_light));
            aim.dir = to_light * (1.0f / aim.dist);
            aim.ndl = dot(aim.dir, st.N);
            pathAimShadowRay(s, st, aim);
            RTX_COUNT(CNT_SHADOW);
            const V3 inv = safeInverse(st.dir);
            float reach;
            gated = sphereTreeGate(s, q, st.orig, inv, st.t_exit, reach);
            minDist = kFltMax, index = -1;
            state = kWalked;
            if (gated) {
                w = sphereWalkerStart(s, st.orig, st.dir, inv, reach, minDist);
                state = kWalking;
            }
        }
        if (__ballot(state != kIdle) == 0ull) break;  // nothing under way, and nothing was left to take
        // ---- the walks, until too few lanes are left walking (all of them once the list is used up) ----
        sphereWalkSome<STATS>(s, w, st.orig, st.dir, st.t_exit, minDist, index, stack, kBlock, more ? kWfPoolLow : 1u, lc);
        if (state == kWalking && !sphereWalkerLive(w)) state = kWalked;
    }
    flushCounters<STATS>(lc, counters);
}

template <bool STATS, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_wf_shade(DevScene s, DevFrame fr, WfArgs a, Counters *counters) {
    __shared__ SmallScene q;
    __shared__ WfChunks ch;
    loadSmallScene(q, s);
    const int level = a.level, depth = fr.depth - level;
    wfChunksLoad(ch, a.nodes);
    const unsigned n_chunks = ch.n_chunks;
    LaneCounters<STATS> lc;
    // (a node costs about the same everywhere: the chunks are dealt out statically)
    for (unsigned chunk = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6); chunk < n_chunks; chunk += gridDim.x * (kBlock / 64u)) {
        unsigned i;
        const bool valid = wfChunkEntry(ch, a.nodes, chunk, i);
        bool doRefl = false, doRefr = false;
        V3 reflO = v3(0, 0, 0), reflD = v3(0, 0, 1), refrO = v3(0, 0, 0), refrD = v3(0, 0, 1);
        if (valid) {
            const float4 f0 = a.pt[i], f1 = a.nrm[i], f2 = a.vw[i], f3 = a.dif[i];
            PathState st;
            st.point = v3(f0.x, f0.y, f0.z), st.N = v3(f1.x, f1.y, f1.z), st.view = v3(f2.x, f2.y, f2.z);
            st.m = materialAt<WIDE>(s, q, __float_as_int(f0.w));
            st.m.dr = f3.x, st.m.dg = f3.y, st.m.db = f3.z;
            st.Ld = 0.0f, st.Ls = 0.0f;
            const unsigned lit = __float_as_uint(f1.w) & ~a.hidden[i];  // cast and not hidden
            const unsigned link = __float_as_uint(f3.w);
            // calcLightning's sums in light order, src/Scene.cpp:73-78
            for (int li = 0; li < s.n_lights; ++li) {
                if (!((lit >> li) & 1u)) continue;
                RTX_COUNT(CNT_LIT);
                const float4 L = lightAt<WIDE>(s, q, li);
                const V3 lightDir = normalize(v3(L.x, L.y, L.z) - st.point);
                pathAddLight(q, st, lightDir, dot(lightDir, st.N), L.w);
            }
            // the node's own colour and its children, :95-105 (see pathConsume)
            const Material &m = st.m;
            const V3 dir = st.view;
            const V3 base = v3(m.dr * st.Ld * m.a0, m.dg * st.Ld * m.a0, m.db * st.Ld * m.a0) + v3(st.Ls * m.a1, st.Ls * m.a1, st.Ls * m.a1);
            doRefl = !s.prune || m.a2 != 0.0f;
            doRefr = !s.prune || m.a3 != 0.0f;
            const auto childOrigin = [&](V3 d) {
#pragma clang fp contract(off)
                return st.point + st.N * (__builtin_signbit(dot(d, st.N)) ? -s.fprec : s.fprec);
            };
            if (doRefl) {
                reflD = reflectDir(dir, st.N);
                reflO = childOrigin(reflD);
            }
            if (doRefr) {
                refrD = normalize(refractDir(dir, st.N, m.refractive));
                refrO = childOrigin(refrD);
            }
            if (depth == 1) {
                // the children have depth 0: the sky colour of their direction, in place (:87-88)
                V3 c = base;
                if (doRefl) {
                    RTX_COUNT(CNT_SKY);
                    c = c + skyColor(s, reflD) * m.a2;
                }
                if (doRefr) {
                    RTX_COUNT(CNT_SKY);
                    c = c + skyColor(s, refrD) * m.a3;
                }
                wfDeliver(a.parent, a.parent_n, level, link, c);
                doRefl = doRefr = false;
            } else {
                a.thin[i] = make_float4(base.x, base.y, base.z, m.a2);
                wfThinMeta(a.thin, a.thin_n)[i] =
                    make_uint2(__float_as_uint(m.a3), (link & (kLinkIndex | kLinkRefr)) | (doRefl ? kHasRefl : 0u) | (doRefr ? kHasRefr : 0u));
            }
        }
        // the children's requests, classed by their rays
        if (__ballot(doRefl || doRefr)) {
            const unsigned at_refl = wfAppendClassed(a.next_req, chunk, doRefl, wfMeshClass(s, reflO, reflD));
            const unsigned at_refr = wfAppendClassed(a.next_req, chunk, doRefr, wfMeshClass(s, refrO, refrD));
            if (doRefl) {
                a.next_o[at_refl] = make_float4(reflO.x, reflO.y, reflO.z, __uint_as_float(i));
                a.next_d[at_refl] = make_float4(reflD.x, reflD.y, reflD.z, 0.0f);
            }
            if (doRefr) {
                a.next_o[at_refr] = make_float4(refrO.x, refrO.y, refrO.z, __uint_as_float(i | kLinkRefr));
                a.next_d[at_refr] = make_float4(refrD.x, refrD.y, refrD.z, 0.0f);
            }
        }
    }
    flushCounters<STATS>(lc, counters);
}

// bottom-up: (own + reflect * a2) + refract * a3, src/Scene.cpp:103-105, into the parent's record
__global__ __launch_bounds__(kBlock) void k_wf_resolve(WfArgs a) {
    __shared__ WfChunks ch;
    const int level = a.level;
    wfChunksLoad(ch, a.nodes);
    const unsigned n_chunks = ch.n_chunks;
    for (unsigned chunk = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6); chunk < n_chunks; chunk += gridDim.x * (kBlock / 64u)) {
        unsigned i;
        if (!wfChunkEntry(ch, a.nodes, chunk, i)) continue;
        const float4 own = a.thin[i];
        const uint2 meta = wfThinMeta(a.thin, a.thin_n)[i];
        const unsigned link = meta.y;
        V3 c = v3(own.x, own.y, own.z);
        if (link & kHasRefl) {
            const float4 kid = a.thin[(size_t)a.thin_n + i];
            c = c + v3(kid.x, kid.y, kid.z) * own.w;
        }
        if (link & kHasRefr) {
            const float4 kid = a.thin[2 * (size_t)a.thin_n + i];
            c = c + v3(kid.x, kid.y, kid.z) * __uint_as_float(meta.x);
        }
        wfDeliver(a.parent, a.parent_n, level, link, c);
    }
}

// The strip's sample colours -> B,G,R bytes (getRGBColor, src/Scene.cpp:204-210); one wave per tile, the
// tile's bytes collected in LDS and stored as rows of dwords where the frame allows.
__global__ __launch_bounds__(kBlock) void k_wf_pixels(DevFrame fr, WfStrip strip, const float4 *__restrict__ samples,
                                                      unsigned char *__restrict__ out) {
    __shared__ unsigned char s_tile[kBlock / 64][192];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool dword_rows = (fr.width & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0;
    for (unsigned tl = blockIdx.x * (kBlock / 64u) + wave; tl < strip.n_tiles; tl += gridDim.x * (kBlock / 64u)) {
        const unsigned tile = strip.first_tile + tl;
        const int x0 = (int)(tile % strip.tiles_x) * 8, y0 = (int)(tile / strip.tiles_x) * 8;
        const int px = x0 + (int)(lane & 7u), pk = y0 + (int)(lane >> 3);
        const bool inside = px < fr.width && pk < fr.rows;
        if (inside) {
            const float4 *sp = samples + ((size_t)tl * 64u + lane) * (unsigned)fr.spp;
            V3 c = v3(sp[0].x, sp[0].y, sp[0].z);
            if (fr.spp > 1) {  // (((c0 + c1) + c2) + c3) * 0.25
                for (int k = 1; k < fr.spp; ++k) c = c + v3(sp[k].x, sp[k].y, sp[k].z);
                c = c * 0.25f;
            }
            unsigned char *bgr = &s_tile[wave][3 * lane];
            bgr[0] = toByte(c.z), bgr[1] = toByte(c.y), bgr[2] = toByte(c.x);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (dword_rows && x0 + 8 <= fr.width && y0 + 8 <= fr.rows) {
            if (lane < 48u) {
                const unsigned r = lane / 6u, w = lane % 6u;
                unsigned *dst = reinterpret_cast<unsigned *>(out + ((size_t)(y0 + (int)r) * fr.width + x0) * 3) + w;
                *dst = *reinterpret_cast<const unsigned *>(&s_tile[wave][4 * lane]);
            }
        } else if (inside) {
            const unsigned char *src = &s_tile[wave][3 * lane];
            unsigned char *p = out + ((size_t)pk * fr.width + px) * 3;
            p[0] = src[0], p[1] = src[1], p[2] = src[2];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
