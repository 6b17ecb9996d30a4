// shotgraph.hip -- the host state machine of one shot, array in / array out (no device work, no Python objects):
//   pvf_lane_*      one pass over one shot                     reference pyannote/video/tracking.py:184-259 (_track)
//   pvf_shot_tracks the shot's graph -> its tracks             :261-357 (_fix, _fill_gaps, connected components, the (min_t, max_t) sort)
//   pvf_track_rows  the track file's numbers of those tracks   scripts/pyannote-face.py:125-127,142-145,262-266
//   pvf_round_decimals  round(float, k) for arrays
// Semantics are those of pyannote_video_amd/tracking_by_detection.py (the Python form, executed against the reference file itself by
// tests/refhost.py): same node / edge order, same association (pvf_associate), same summation order in _fix, same tie rules of every
// sort.  tests/test_shotgraph.py runs both forms on thousands of scripted shots and compares every track row.
//
// A lane is a RESUMABLE state machine: the trackers' starts and first updates were issued ahead in bulk (the "plan"), so a pass runs
// on arrays alone until a tracker outlives its first update -- then the lane returns a request (commit the deferred filter update,
// update on the current frame, or the next window of the plan) and is advanced again with the reply.  The engine's tracking thread
// spent 16 + 6 + 7 ms per 250-frame shot in the Python form of these three steps, the last shot's on the
// critical path of a step.
#include "pvf_internal.h"
#include <algorithm>
#include <cfenv>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>

#define API_BEGIN try {
#define API_END                                                        \
    return 0;                                                          \
    }                                                                  \
    catch (const std::exception& e) { pvf_set_error(e.what()); return -1; } \
    catch (...) { pvf_set_error("unknown error"); return -2; }

namespace {
enum { KIND_FWD = 1, KIND_DET = 2, KIND_BWD = 3 };          // the reference's _STATUS_ORDER (forward 1, detection 2, backward 3)
enum { REQ_DONE = 0, REQ_UPDATE = 1, REQ_COMMIT = 2, REQ_PLAN = 3 };

struct Node { int32_t f; int32_t kind; double box[4]; };

struct Lane {
    int n = 0, dir = KIND_FWD;
    double min_conf = 0, ratio = 0;
    bool deferring = true;
    std::vector<int32_t> counts, starts;                    // detections per shot frame (forward index), prefix sums
    std::vector<double> boxes;                              // [sum][4]
    // the plan, per PROCESSING frame p: where its trackers sit in plan_* (-1: not fed yet), whether they have a first update
    std::vector<int64_t> plan_at;
    std::vector<uint8_t> plan_has;
    std::vector<uint64_t> plan_h;
    std::vector<double> plan_psr, plan_pos;
    // state
    int p = 0, phase = 0;
    std::vector<int> active, ids, need, late;
    std::vector<uint64_t> handle;
    std::vector<double> pos, conf, cpsr, cpos, fresh_psr, fresh_pos;
    std::vector<uint8_t> cached;
    std::vector<int32_t> uncommitted;
    std::vector<Node> prev;
    std::vector<Node> eu, ev;                               // edges in the order the reference calls add_edge
    std::vector<double> econf;
    std::vector<uint64_t> dead;                             // killed trackers not yet handed to the caller
    bool finished = false;
    int frame_of(int pp) const { return dir == KIND_BWD ? n - 1 - pp : pp; }
};

std::mutex g_mu;
std::unordered_map<uint64_t, std::unique_ptr<Lane>> g_lanes;
uint64_t g_next = 0x5000;

Lane* lane_of(pvf_handle h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_lanes.find(h);
    PVF_REQUIRE(it != g_lanes.end(), "unknown lane handle");
    return it->second.get();
}

void kill(Lane& L, int id)
{
    L.active.erase(std::find(L.active.begin(), L.active.end(), id));
    L.cached[id] = 0;
    L.uncommitted[id] = -1;
    L.dead.push_back(L.handle[id]);
}

void add_edge(Lane& L, const Node& u, const Node& v, double c)
{
    L.eu.push_back(u); L.ev.push_back(v); L.econf.push_back(c);
}

// one step of the lane; returns the request kind
int advance(Lane& L, const double* rpsr, const double* rpos, int n_reply, std::vector<uint64_t>& req_h, std::vector<int32_t>& req_f, int* plan_from)
{
    req_h.clear(); req_f.clear();
    for (;;) {
        switch (L.phase) {
        case 0: {
            if (L.p >= L.n) {
                while (!L.active.empty()) kill(L, L.active.front());
                L.finished = true;
                return REQ_DONE;
            }
            L.ids = L.active;
            L.need.clear(); L.late.clear();
            for (int id : L.ids) if (!L.cached[id]) L.need.push_back(id);
            for (int id : L.need) if (L.uncommitted[id] >= 0) L.late.push_back(id);
            L.phase = 1;
            if (!L.late.empty()) {
                // survivors of a deferred first update: their filters are brought up to date (on the frame that update ran on) before the next update
                for (int id : L.late) { req_h.push_back(L.handle[id]); req_f.push_back(L.uncommitted[id]); L.uncommitted[id] = -1; }
                return REQ_COMMIT;
            }
            break;
        }
        case 1: {
            if (!L.need.empty()) {
                const int f = L.frame_of(L.p);
                for (int id : L.need) { req_h.push_back(L.handle[id]); req_f.push_back(f); }
                L.phase = 2;
                return REQ_UPDATE;
            }
            L.phase = 3;
            break;
        }
        case 2: {
            PVF_REQUIRE(n_reply == (int)L.need.size() && rpsr && rpos, "pvf_lane_advance: the reply does not match the update request");
            L.fresh_psr.assign(rpsr, rpsr + n_reply);
            L.fresh_pos.assign(rpos, rpos + 4 * (size_t)n_reply);
            L.phase = 3;
            break;
        }
        case 3: {
            const int f = L.frame_of(L.p);
            size_t k = 0;
            for (int id : L.ids) {
                double c; const double* q;
                if (L.cached[id]) {
                    if (L.deferring) L.uncommitted[id] = f;
                    c = L.cpsr[id]; q = &L.cpos[4 * (size_t)id];
                    L.cached[id] = 0;
                } else {
                    PVF_REQUIRE(k < L.need.size() && L.need[k] == id, "lane: update replies out of step");
                    c = L.fresh_psr[k]; q = &L.fresh_pos[4 * k];
                    ++k;
                }
                L.conf[id] = c;
                memcpy(&L.pos[4 * (size_t)id], q, 4 * sizeof(double));
                if (c < L.min_conf) kill(L, id);
            }
            const int m = L.counts[f];
            const double* det = L.boxes.data() + 4 * (size_t)L.starts[f];
            if (!L.active.empty() && m > 0) {
                const int nt = (int)L.active.size();
                std::vector<double> tp(4 * (size_t)nt);
                for (int t = 0; t < nt; ++t) memcpy(&tp[4 * (size_t)t], &L.pos[4 * (size_t)L.active[t]], 4 * sizeof(double));
                std::vector<int32_t> d_of(nt, -1);
                PVF_REQUIRE(pvf_associate(tp.data(), nt, det, m, L.ratio, d_of.data()) == 0, "lane: association failed");
                const std::vector<int> order = L.active;
                for (int t = 0; t < nt; ++t) {
                    if (d_of[t] < 0) continue;
                    const int id = order[t];
                    Node cur; cur.f = f; cur.kind = KIND_DET; memcpy(cur.box, det + 4 * (size_t)d_of[t], sizeof cur.box);
                    add_edge(L, L.prev[id], cur, L.conf[id]);
                    kill(L, id);
                }
            }
            for (int id : L.active) {
                Node cur; cur.f = f; cur.kind = L.dir; memcpy(cur.box, &L.pos[4 * (size_t)id], sizeof cur.box);
                add_edge(L, L.prev[id], cur, L.conf[id]);
                L.prev[id] = cur;
            }
            L.phase = 4;
            if (m > 0 && L.plan_at[L.p] < 0) { *plan_from = L.p; return REQ_PLAN; }
            break;
        }
        case 4: {
            const int f = L.frame_of(L.p);
            const int m = L.counts[f];
            if (m > 0) {
                PVF_REQUIRE(L.plan_at[L.p] >= 0, "lane: advanced without the plan it asked for");
                const size_t k0 = (size_t)L.plan_at[L.p];
                const double* det = L.boxes.data() + 4 * (size_t)L.starts[f];
                for (int d = 0; d < m; ++d) {
                    const int id = (int)L.handle.size();
                    L.handle.push_back(L.plan_h[k0 + d]);
                    L.pos.resize(4 * (size_t)(id + 1)); L.cpos.resize(4 * (size_t)(id + 1));
                    L.conf.push_back(0); L.cpsr.push_back(0); L.cached.push_back(0); L.uncommitted.push_back(-1);
                    Node s; s.f = f; s.kind = KIND_DET; memcpy(s.box, det + 4 * (size_t)d, sizeof s.box);
                    L.prev.push_back(s);
                    if (L.plan_has[L.p]) {
                        L.cached[id] = 1;
                        L.cpsr[id] = L.plan_psr[k0 + d];
                        memcpy(&L.cpos[4 * (size_t)id], &L.plan_pos[4 * (k0 + d)], 4 * sizeof(double));
                    }
                    L.active.push_back(id);
                }
            }
            ++L.p;
            L.phase = 0;
            break;
        }
        default:
            PVF_REQUIRE(false, "lane: bad state");
        }
    }
}

// ---- the shot's graph -> tracks ---------------------------------------------------------------------------------------------------
struct Key {
    int32_t f, kind; uint64_t b[4];
    bool operator==(const Key& o) const { return f == o.f && kind == o.kind && b[0] == o.b[0] && b[1] == o.b[1] && b[2] == o.b[2] && b[3] == o.b[3]; }
};
struct KeyHash {
    size_t operator()(const Key& k) const
    {
        uint64_t h = 1469598103934665603ull ^ (uint64_t)(uint32_t)k.f * 0x9E3779B97F4A7C15ull ^ ((uint64_t)k.kind << 56);
        for (int i = 0; i < 4; ++i) { h ^= k.b[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); }
        return (size_t)h;
    }
};
Key key_of(const Node& n)
{
    Key k; k.f = n.f; k.kind = n.kind;
    for (int i = 0; i < 4; ++i) { const double v = n.box[i] + 0.0; memcpy(&k.b[i], &v, 8); }     // (-0.0 + 0.0 = +0.0: Python's 0.0 == -0.0)
    return k;
}

struct Row { int32_t f; int32_t box[4]; int32_t status; };          // status: forwards | detections << 8 | backwards << 16 | error << 24

bool no_match(const double* a, const double* b, double ratio)
{
    double v;
    overlap_matrix_host(a, 1, b, 1, ratio, &v);
    return v == 0.0;
}

int32_t round_half_even(double v) { return (int32_t)std::nearbyint(v); }      // Python's round() of a float (FE_TONEAREST is set by the callers)

struct RoundingMode {
    int old;
    RoundingMode() : old(fegetround()) { fesetround(FE_TONEAREST); }
    ~RoundingMode() { fesetround(old); }
};

// alphabetical rank of the status strings (sorted(track) compares them last)
int alpha_rank(int kind) { return kind == KIND_BWD ? 0 : (kind == KIND_DET ? 1 : 2); }

void fix_track(std::vector<Node>& nodes, double ratio, std::vector<Row>& out)
{
    std::sort(nodes.begin(), nodes.end(), [](const Node& a, const Node& b) {
        if (a.f != b.f) return a.f < b.f;
        for (int i = 0; i < 4; ++i) if (a.box[i] != b.box[i]) return a.box[i] < b.box[i];
        return alpha_rank(a.kind) < alpha_rank(b.kind);
    });
    size_t i = 0;
    while (i < nodes.size()) {
        size_t j = i;
        while (j < nodes.size() && nodes[j].f == nodes[i].f) ++j;
        Row r; r.f = nodes[i].f;
        int cnt[4] = {0, 0, 0, 0};
        for (size_t k = i; k < j; ++k) ++cnt[nodes[k].kind];
        bool err = false;
        if (j - i > 1)
            for (size_t a = i; a < j && !err; ++a)
                for (size_t b = a + 1; b < j && !err; ++b) err = no_match(nodes[a].box, nodes[b].box, ratio);
        for (int c = 0; c < 4; ++c) {
            double s = nodes[i].box[c];
            for (size_t k = i + 1; k < j; ++k) s = s + nodes[k].box[c];
            r.box[c] = round_half_even(j - i == 1 ? s : s / (double)(j - i));
        }
        PVF_REQUIRE(cnt[KIND_FWD] < 256 && cnt[KIND_DET] < 256 && cnt[KIND_BWD] < 256, "shot tracks: more than 255 nodes of one kind at one timestamp of a track");
        r.status = cnt[KIND_FWD] | (cnt[KIND_DET] << 8) | (cnt[KIND_BWD] << 16) | ((err ? 1 : 0) << 24);
        out.push_back(r);
        i = j;
    }
}

struct Track { std::vector<Row> rows; int32_t tmin, tmax; };
void span(Track& t)
{
    t.tmin = t.rows[0].f; t.tmax = t.rows[0].f;
    for (const Row& r : t.rows) { t.tmin = std::min(t.tmin, r.f); t.tmax = std::max(t.tmax, r.f); }
}
bool by_span(const Track& a, const Track& b) { return a.tmin != b.tmin ? a.tmin < b.tmin : a.tmax < b.tmax; }

int find_root(std::vector<int>& parent, int i)
{
    while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
    return i;
}
} // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
extern "C" int32_t pvf_lane_create(int32_t n_frames, const int32_t* det_counts, const double* det_boxes, int32_t direction, double min_confidence,
                                   double min_overlap_ratio, int32_t deferring, pvf_handle* out)
{
    API_BEGIN
    PVF_REQUIRE(n_frames >= 0 && (det_counts || n_frames == 0) && out && (direction == KIND_FWD || direction == KIND_BWD), "pvf_lane_create: bad arguments");
    std::unique_ptr<Lane> L(new Lane());
    L->n = n_frames; L->dir = direction; L->min_conf = min_confidence; L->ratio = min_overlap_ratio; L->deferring = deferring != 0;
    L->counts.assign(det_counts, det_counts + n_frames);
    L->starts.resize((size_t)n_frames + 1, 0);
    for (int i = 0; i < n_frames; ++i) { PVF_REQUIRE(det_counts[i] >= 0, "pvf_lane_create: negative count"); L->starts[i + 1] = L->starts[i] + det_counts[i]; }
    const size_t total = (size_t)L->starts[n_frames];
    PVF_REQUIRE(det_boxes || total == 0, "pvf_lane_create: no boxes");
    if (total) L->boxes.assign(det_boxes, det_boxes + 4 * total);
    L->plan_at.assign((size_t)n_frames, -1);
    L->plan_has.assign((size_t)n_frames, 0);
    std::lock_guard<std::mutex> lk(g_mu);
    const uint64_t h = g_next++;
    g_lanes[h] = std::move(L);
    *out = h;
    API_END
}

extern "C" int32_t pvf_lane_destroy(pvf_handle lane)
{
    API_BEGIN
    std::lock_guard<std::mutex> lk(g_mu);
    g_lanes.erase(lane);
    API_END
}

// the plan of the processing frames [p0, p0 + n_p): per frame whether its trackers have a first update (the pass's last frame has none),
// and for the detections of those frames -- frame after frame in PROCESSING order, a frame's detections in their own order -- the tracker
// handle, the first update's confidence and position
extern "C" int32_t pvf_lane_feed_plan(pvf_handle lane, int32_t p0, int32_t n_p, const uint8_t* has_update, const uint64_t* handles, const double* psr,
                                      const double* pos)
{
    API_BEGIN
    Lane& L = *lane_of(lane);
    PVF_REQUIRE(p0 >= 0 && n_p >= 0 && p0 + n_p <= L.n && (has_update || n_p == 0), "pvf_lane_feed_plan: bad range");
    size_t k = 0;
    for (int p = p0; p < p0 + n_p; ++p) {
        const int m = L.counts[L.frame_of(p)];
        L.plan_at[p] = (int64_t)L.plan_h.size();
        L.plan_has[p] = has_update[p - p0];
        if (m == 0) continue;
        PVF_REQUIRE(handles && (!has_update[p - p0] || (psr && pos)), "pvf_lane_feed_plan: missing arrays");
        for (int d = 0; d < m; ++d, ++k) {
            L.plan_h.push_back(handles[k]);
            L.plan_psr.push_back(has_update[p - p0] ? psr[k] : 0.0);
            for (int c = 0; c < 4; ++c) L.plan_pos.push_back(has_update[p - p0] ? pos[4 * k + c] : 0.0);
        }
    }
    API_END
}

// request: 0 = the pass is over, 1 = update req_handles on the frames req_frames (shot frame indices; reply with their confidences and
// positions in the next call), 2 = commit the deferred update of req_handles on req_frames (no reply data), 3 = feed the plan from
// processing frame *plan_from on (no reply data).  cap: room of the request arrays (the shot's number of detections always suffices).
extern "C" int32_t pvf_lane_advance(pvf_handle lane, const double* reply_psr, const double* reply_pos, int32_t n_reply, int32_t* request,
                                    uint64_t* req_handles, int32_t* req_frames, int32_t cap, int32_t* n_req, int32_t* plan_from)
{
    API_BEGIN
    Lane& L = *lane_of(lane);
    PVF_REQUIRE(request && n_req && plan_from && cap >= 0, "pvf_lane_advance: bad arguments");
    PVF_REQUIRE(!L.finished, "pvf_lane_advance: the pass is over");
    std::vector<uint64_t> rh; std::vector<int32_t> rf;
    int pf = -1;
    *request = advance(L, reply_psr, reply_pos, n_reply, rh, rf, &pf);
    // (a request names trackers that are alive: never more than the shot has detections -- the caller sizes its arrays by that)
    PVF_REQUIRE((int)rh.size() <= cap && (rh.empty() || (req_handles && req_frames)), "pvf_lane_advance: request arrays too small");
    *n_req = (int32_t)rh.size();
    *plan_from = pf;
    for (size_t i = 0; i < rh.size(); ++i) { req_handles[i] = rh[i]; req_frames[i] = rf[i]; }
    API_END
}

// the trackers killed since the last call, in the order they were killed (the caller releases them); *n > cap: call again with room
extern "C" int32_t pvf_lane_take_dead(pvf_handle lane, uint64_t* out, int32_t cap, int32_t* n)
{
    API_BEGIN
    Lane& L = *lane_of(lane);
    PVF_REQUIRE(n && cap >= 0, "pvf_lane_take_dead: bad arguments");
    *n = (int32_t)L.dead.size();
    if (*n > cap) return 0;
    for (size_t i = 0; i < L.dead.size(); ++i) out[i] = L.dead[i];
    L.dead.clear();
    API_END
}

// edges of a finished (or running) pass for the reference's own data structure (finish_shot_graph): node = (frame, kind, box)
extern "C" int32_t pvf_lane_edges(pvf_handle lane, int32_t* n_edges, int32_t* u_frame_kind, double* u_box, int32_t* v_frame_kind, double* v_box, double* conf,
                                  int32_t cap)
{
    API_BEGIN
    Lane& L = *lane_of(lane);
    PVF_REQUIRE(n_edges, "pvf_lane_edges: bad arguments");
    *n_edges = (int32_t)L.eu.size();
    if (!u_frame_kind || (int)L.eu.size() > cap) return 0;
    for (size_t k = 0; k < L.eu.size(); ++k) {
        u_frame_kind[2 * k] = L.eu[k].f; u_frame_kind[2 * k + 1] = L.eu[k].kind; memcpy(u_box + 4 * k, L.eu[k].box, 32);
        v_frame_kind[2 * k] = L.ev[k].f; v_frame_kind[2 * k + 1] = L.ev[k].kind; memcpy(v_box + 4 * k, L.ev[k].box, 32);
        conf[k] = L.econf[k];
    }
    API_END
}

// tracks of a shot from its two finished passes: rows [cap][6] = (frame, l, t, r, b, status code), track k = rows track_start[k] ..
// track_start[k + 1]; *n_rows > cap or *n_tracks > track_cap: nothing written, call again with room.
// status code: forwards | detections << 8 | backwards << 16 | error << 24 (the reference's "+".join in _STATUS_ORDER, "error(...)")
extern "C" int32_t pvf_shot_tracks(pvf_handle lane_forward, pvf_handle lane_backward, const double* times, int32_t n_frames, double max_gap,
                                   int32_t* rows, int32_t cap, int32_t* n_rows, int32_t* track_start, int32_t track_cap, int32_t* n_tracks)
{
    API_BEGIN
    RoundingMode rm;
    Lane& F = *lane_of(lane_forward);
    Lane& B = *lane_of(lane_backward);
    PVF_REQUIRE(times && n_rows && n_tracks && F.n == n_frames && B.n == n_frames && F.finished && B.finished, "pvf_shot_tracks: two finished passes of one shot");
    const double ratio = F.ratio;
    // node order of the reference's graph: the detections frame by frame (a box seen twice on a frame is one node), then the nodes the
    // forward edges name (u, then v), then the backward ones
    std::unordered_map<Key, int, KeyHash> index;
    std::vector<Node> nodes;
    std::vector<int> parent;
    auto node_id = [&](const Node& n) {
        const Key k = key_of(n);
        auto it = index.find(k);
        if (it != index.end()) return it->second;
        const int id = (int)nodes.size();
        index.emplace(k, id);
        nodes.push_back(n);
        parent.push_back(id);
        return id;
    };
    for (int f = 0; f < n_frames; ++f)
        for (int d = 0; d < F.counts[f]; ++d) {
            Node n; n.f = f; n.kind = KIND_DET; memcpy(n.box, F.boxes.data() + 4 * ((size_t)F.starts[f] + d), sizeof n.box);
            node_id(n);
        }
    for (Lane* L : {&F, &B})
        for (size_t k = 0; k < L->eu.size(); ++k) {
            int iu = node_id(L->eu[k]), iv = node_id(L->ev[k]);
            iu = find_root(parent, iu); iv = find_root(parent, iv);
            if (iu != iv) { if (iu < iv) parent[iv] = iu; else parent[iu] = iv; }
        }
    // components in the order of their first node, their nodes in node order
    std::vector<int> comp_of(nodes.size(), -1);
    std::vector<std::vector<Node>> comps;
    for (size_t i = 0; i < nodes.size(); ++i) {
        const int r = find_root(parent, (int)i);
        if (comp_of[r] < 0) { comp_of[r] = (int)comps.size(); comps.emplace_back(); }
        comps[comp_of[r]].push_back(nodes[i]);
    }
    std::vector<Track> tracks(comps.size());
    for (size_t k = 0; k < comps.size(); ++k) { fix_track(comps[k], ratio, tracks[k].rows); span(tracks[k]); }
    // _fill_gaps: tracks in (min_t, max_t) order; an edge i < j when j starts at most max_gap after i ends and their end / start boxes match
    std::stable_sort(tracks.begin(), tracks.end(), by_span);
    const int T = (int)tracks.size();
    std::vector<int> cp(T);
    for (int i = 0; i < T; ++i) cp[i] = i;
    for (int i = 0; i < T; ++i)
        for (int j = i + 1; j < T; ++j) {
            const Row& a = tracks[i].rows.back();
            const Row& b = tracks[j].rows.front();
            const double ti = times[a.f], tj = times[b.f];
            if (tj < ti || tj - ti > max_gap) continue;
            const double ba[4] = {(double)a.box[0], (double)a.box[1], (double)a.box[2], (double)a.box[3]};
            const double bb[4] = {(double)b.box[0], (double)b.box[1], (double)b.box[2], (double)b.box[3]};
            if (!no_match(ba, bb, ratio)) {
                const int ri = find_root(cp, i), rj = find_root(cp, j);
                if (ri != rj) { if (ri < rj) cp[rj] = ri; else cp[ri] = rj; }
            }
        }
    std::vector<Track> merged;
    std::vector<int> slot(T, -1);
    for (int i = 0; i < T; ++i) {                            // components by their smallest member, members ascending
        const int r = find_root(cp, i);
        if (slot[r] < 0) { slot[r] = (int)merged.size(); merged.emplace_back(); }
        Track& m = merged[slot[r]];
        m.rows.insert(m.rows.end(), tracks[i].rows.begin(), tracks[i].rows.end());
    }
    for (Track& m : merged) span(m);
    std::stable_sort(merged.begin(), merged.end(), by_span);
    size_t total = 0;
    for (const Track& m : merged) total += m.rows.size();
    *n_rows = (int32_t)total;
    *n_tracks = (int32_t)merged.size();
    if ((int64_t)total > cap || (int)merged.size() > track_cap || !rows || !track_start) return 0;
    size_t o = 0;
    for (size_t k = 0; k < merged.size(); ++k) {
        track_start[k] = (int32_t)o;
        for (const Row& r : merged[k].rows) {
            int32_t* q = rows + 6 * o++;
            q[0] = r.f; q[1] = r.box[0]; q[2] = r.box[1]; q[3] = r.box[2]; q[4] = r.box[3]; q[5] = r.status;
        }
    }
    track_start[merged.size()] = (int32_t)o;
    API_END
}

// round(x, decimals) of Python floats for an array: the double nearest to the decimal number nearest to x (ties between decimals to
// even; a tie is decided on the EXACT binary value of x, as the float formatting behind Python's round() does).  |x| < 2^40, decimals 0..6.
static double round_decimal(double x, int decimals)
{
    static const double P10[7] = {1.0, 10.0, 100.0, 1000.0, 10000.0, 100000.0, 1000000.0};
    if (!(std::fabs(x) < 1099511627776.0)) return x;         // (inf, nan, huge: unchanged -- not produced by this path)
    const bool neg = std::signbit(x);
    const double a = std::fabs(x);
    int e;
    const double m = std::frexp(a, &e);                      // a = m * 2^e, 0.5 <= m < 1
    if (a == 0.0) return x;
    const unsigned __int128 M = (unsigned __int128)(uint64_t)std::ldexp(m, 53);      // a = M * 2^(e - 53) exactly
    const int sh = 53 - e;                                   // a * 10^k = M * 10^k / 2^sh
    const unsigned __int128 num = M * (unsigned __int128)(uint64_t)P10[decimals];
    unsigned __int128 q, rem, half;
    if (sh <= 0) { q = num << (-sh); rem = 0; half = 1; }
    else if (sh >= 120) { q = 0; rem = num; half = (unsigned __int128)1 << 119; if (sh > 120) { rem = 0; } }
    else { q = num >> sh; rem = num & (((unsigned __int128)1 << sh) - 1); half = (unsigned __int128)1 << (sh - 1); }
    if (sh > 0 && sh < 120) {
        if (rem > half || (rem == half && (q & 1))) ++q;
    } else if (sh >= 120) {
        q = 0;                                               // a * 10^k < 2^-60: rounds to zero
    }
    const double r = (double)(uint64_t)q / P10[decimals];    // both exact doubles: the quotient is the correctly rounded decimal
    return neg ? -r : r;
}

extern "C" int32_t pvf_round_decimals(const double* in, int64_t n, int32_t decimals, double* out)
{
    API_BEGIN
    PVF_REQUIRE(n >= 0 && (n == 0 || (in && out)) && decimals >= 0 && decimals <= 6, "pvf_round_decimals: bad arguments");
    RoundingMode rm;
    for (int64_t i = 0; i < n; ++i) out[i] = round_decimal(in[i], decimals);
    API_END
}

// the numbers of the track file and of `extract` for n track rows (reference scripts/pyannote-face.py:262-266 writes '%.3f' of box / frame
// size; :125-127 reads them back as float32; :142-145 multiplies by the frame size and truncates):
//   file_box[n][4]  = float64(float32(round(box / (width, height, width, height), 3)))
//   pixel_box[n][4] = int(file_box * (width, height, width, height))
extern "C" int32_t pvf_track_rows(const int32_t* boxes, int64_t n, int32_t det_width, int32_t det_height, int32_t width, int32_t height, double* file_box,
                                  int32_t* pixel_box)
{
    API_BEGIN
    PVF_REQUIRE(n >= 0 && (n == 0 || (boxes && file_box && pixel_box)) && det_width > 0 && det_height > 0 && width > 0 && height > 0, "pvf_track_rows: bad arguments");
    RoundingMode rm;
    const double den[4] = {(double)det_width, (double)det_height, (double)det_width, (double)det_height};
    const double mul[4] = {(double)width, (double)height, (double)width, (double)height};
    for (int64_t i = 0; i < 4 * n; ++i) {
        const double v = (double)boxes[i] / den[i & 3];
        const double q = (double)(float)round_decimal(v, 3);
        file_box[i] = q;
        pixel_box[i] = (int32_t)(q * mul[i & 3]);              // int(): truncation
    }
    API_END
}
