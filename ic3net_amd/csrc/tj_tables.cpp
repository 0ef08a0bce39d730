// tj_tables.cpp — host-side, init-time constant tables of the Traffic-Junction world (uploaded once).
//
// Reproduces, as data, what the reference computes at construction:
//   road-id grid      traffic_junction_env.py:300-319 (_set_grid) over traffic_helper.py:5-22 (road slices)
//   routes            traffic_junction_env.py:395-410 (easy), traffic_helper.py:28-209 (medium/hard: a walk
//                     over the lane/junction maps, first admissible neighbour in (-1,0),(1,0),(0,-1),(0,1) order)
// Pinned against tables captured from the reference (tests/golden/tj_tables.npz) through ic3_tj_get_tables.
#include <array>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "ic3_common.hpp"

namespace ic3 {
namespace {

struct Rect {
    int r0, r1, c0, c1;  // half-open
};

using Cell = std::pair<int, int>;

struct World {
    int h = 0, w = 0;
    std::vector<int> road;      // 1 on road cells (route_grid, traffic_junction_env.py:309)
    std::vector<int> lane;      // road_dir (traffic_helper.py:34,49-52,82-90)
    std::vector<int> junction;  // traffic_helper.py:35,54,93-97
    std::vector<Cell> arrive, finish;
    int at(const std::vector<int>& m, Cell p) const { return m[(size_t)p.first * w + p.second]; }
};

std::vector<Rect> road_rects(int h, int w, int difficulty)
{
    if (difficulty == IC3_TJ_EASY) return { { h / 2, h / 2 + 1, 0, w }, { 0, h, w / 2, w / 2 + 1 } };
    if (difficulty == IC3_TJ_MEDIUM) return { { h / 2 - 1, h / 2 + 1, 0, w }, { 0, h, w / 2 - 1, w / 2 + 1 } };
    // hard; the last column band is computed from h in the reference (traffic_helper.py:20) — dims are square
    return { { h / 3 - 2, h / 3, 0, w }, { 2 * h / 3, 2 * h / 3 + 2, 0, w }, { 0, h, w / 3 - 2, w / 3 },
             { 0, h, 2 * h / 3, 2 * h / 3 + 2 } };
}

void fill_row(std::vector<int>& m, int w, int r, int val)
{
    for (int c = 0; c < w; ++c) m[(size_t)r * w + c] = val;
}
void fill_col(std::vector<int>& m, int h, int w, int c, int val)
{
    for (int r = 0; r < h; ++r) m[(size_t)r * w + c] = val;
}
void fill_box(std::vector<int>& m, int w, int r0, int c0, int val)
{
    for (int r = r0; r < r0 + 2; ++r)
        for (int c = c0; c < c0 + 2; ++c) m[(size_t)r * w + c] = val;
}

void annotate(World& g, int difficulty)
{
    const int h = g.h, w = g.w;
    g.lane = g.road;
    g.junction.assign((size_t)h * w, 0);
    if (difficulty == IC3_TJ_MEDIUM) {
        g.arrive = { { 0, w / 2 - 1 }, { h - 1, w / 2 }, { h / 2, 0 }, { h / 2 - 1, w - 1 } };
        g.finish = { { 0, w / 2 }, { h - 1, w / 2 - 1 }, { h / 2 - 1, 0 }, { h / 2, w - 1 } };
        fill_row(g.lane, w, h / 2, 2);
        fill_row(g.lane, w, h / 2 - 1, 3);
        fill_col(g.lane, h, w, w / 2, 4);
        fill_box(g.junction, w, h / 2 - 1, w / 2 - 1, 1);
    } else {
        g.arrive = { { 0, w / 3 - 2 },     { 0, 2 * w / 3 },         { h / 3 - 1, 0 },     { 2 * h / 3 + 1, 0 },
                     { h - 1, w / 3 - 1 }, { h - 1, 2 * w / 3 + 1 }, { h / 3 - 2, w - 1 }, { 2 * h / 3, w - 1 } };
        g.finish = { { 0, w / 3 - 1 },     { 0, 2 * w / 3 + 1 }, { h / 3 - 2, 0 },     { 2 * h / 3, 0 },
                     { h - 1, w / 3 - 2 }, { h - 1, 2 * w / 3 }, { h / 3 - 1, w - 1 }, { 2 * h / 3 + 1, w - 1 } };
        fill_row(g.lane, w, h / 3 - 1, 2);
        fill_row(g.lane, w, 2 * h / 3, 3);
        fill_row(g.lane, w, 2 * h / 3 + 1, 4);
        fill_col(g.lane, h, w, w / 3 - 2, 5);
        fill_col(g.lane, h, w, w / 3 - 1, 6);
        fill_col(g.lane, h, w, 2 * w / 3, 7);
        fill_col(g.lane, h, w, 2 * w / 3 + 1, 8);
        for (int r0 : { h / 3 - 2, 2 * h / 3 })
            for (int c0 : { w / 3 - 2, 2 * w / 3 }) fill_box(g.junction, w, r0, c0, 1);
    }
}

struct Hop {
    Cell next;
    bool progressed = false, completed = false, ok = false;
};

// traffic_helper.py:99-152 — which neighbour a car at `cur` moves to, given the turn it is executing
Hop hop(const World& g, Cell cur, int turn, int turn_step, Cell origin, const std::set<Cell>& seen)
{
    static const int DR[4] = { -1, 1, 0, 0 }, DC[4] = { 0, 0, -1, 1 };
    Hop out;
    for (int k = 0; k < 4; ++k) {
        const Cell n{ cur.first + DR[k], cur.second + DC[k] };
        if (n.first < 0 || n.first > g.h - 1 || n.second < 0 || n.second > g.w - 1) continue;
        if (!g.at(g.road, n) || seen.count(n)) continue;
        const bool jn = g.at(g.junction, n) == 1, jc = g.at(g.junction, cur) == 1;
        bool take = false, prog = false, done = false;
        if (jn && jc) {
            if ((turn == 0 || turn == 2) && (n.first == origin.first || n.second == origin.second)) {
                take = true;
                prog = (turn == 2);
            } else if (turn == 2 && turn_step == 1) {
                take = true;
                prog = true;
            }
        } else if (jc && !jn && turn == 2 && turn_step == 2 &&
                   (std::abs(origin.first - n.first) == 2 || std::abs(origin.second - n.second) == 2)) {
            take = done = true;
        } else if (jn && !jc) {
            take = true;
        } else if (turn == 1 && !jn && jc) {
            take = done = true;
        } else if (turn == 0 && jc && g.at(g.lane, n) == g.at(g.lane, origin)) {
            take = done = true;
        } else if (g.at(g.lane, n) == g.at(g.lane, cur) && !jc) {
            take = true;
        }
        // the reference accumulates flags over ALL admissible neighbours but moves to the first one
        if (take) {
            if (!out.ok) {
                out.next = n;
                out.ok = true;
            }
            out.progressed |= prog;
            out.completed |= done;
        }
    }
    return out;
}

}  // namespace

int tj_build_tables(int dim, int vision, int difficulty, int* h_out, int* w_out, int* base_out, int* npath_out,
                    int* narrival_out, int* rpa_out, std::vector<int32_t>& grid, std::vector<int32_t>& route_off,
                    std::vector<int32_t>& route_rc, std::string& err, std::vector<int32_t>* road_out)
{
    // traffic_junction_env.py:93-100 config asserts
    if (difficulty == IC3_TJ_EASY || difficulty == IC3_TJ_MEDIUM) {
        if (dim % 2 != 0) { err = "Only even dimension supported for now."; return -22; }
        if (dim < 4 + vision) { err = "Min dim: 4 + vision"; return -22; }
    } else if (difficulty == IC3_TJ_HARD) {
        if (dim < 9) { err = "Min dim: 9"; return -22; }
        if (dim % 3 != 0) { err = "Hard version works for multiple of 3. dim. only."; return -22; }
    } else {
        err = "unknown difficulty";
        return -22;
    }
    World g;
    g.h = g.w = (difficulty == IC3_TJ_EASY) ? dim + 1 : dim;  // :111-115
    const int h = g.h, w = g.w;
    const int nroad = difficulty == IC3_TJ_EASY ? 2 : difficulty == IC3_TJ_MEDIUM ? 4 : 8;           // :117-119
    const int base = (difficulty == IC3_TJ_EASY ? 2 : difficulty == IC3_TJ_MEDIUM ? 4 : 8) * dim;  // :121-124 (original dim)
    const int npath = nroad * (nroad - 1);                                                          // nPr(nroad, 2) :126
    const int outside = base;                                                                       // :131

    grid.assign((size_t)h * w, outside);
    g.road.assign((size_t)h * w, 0);
    int next_id = 0;
    for (const Rect& rc : road_rects(h, w, difficulty)) {  // :306-314, later rects overwrite junction cells
        for (int r = rc.r0; r < rc.r1; ++r)
            for (int c = rc.c0; c < rc.c1; ++c) {
                g.road[(size_t)r * w + c] = 1;
                grid[(size_t)r * w + c] = next_id++;
            }
    }

    std::vector<std::vector<std::vector<Cell>>> routes;
    if (difficulty == IC3_TJ_EASY) {  // :395-410
        std::vector<Cell> top, left;
        for (int i = 0; i < h; ++i) top.push_back({ i, w / 2 });
        for (int i = 0; i < w; ++i) left.push_back({ h / 2, i });
        routes = { { top }, { left } };
    } else {
        annotate(g, difficulty);
        const int second = (difficulty == IC3_TJ_MEDIUM) ? 1 : 3;  // traffic_helper.py:168-169
        for (size_t i = 0; i < g.arrive.size(); ++i) {
            std::vector<std::vector<Cell>> paths;
            for (int t1 = 0; t1 < 3; ++t1) {
                for (int t2 = 0; t2 < second; ++t2) {
                    int nturns = 0, turn = t1, tstep = 0;
                    Cell cur = g.arrive[i], origin = cur;
                    std::vector<Cell> path{ cur };
                    std::set<Cell> seen;
                    auto at_goal = [&](Cell p) {
                        for (size_t k = 0; k < g.finish.size(); ++k)
                            if (k != i && g.finish[k] == p) return true;
                        return false;
                    };
                    int guard = 0;
                    while (!at_goal(cur)) {
                        seen.insert(cur);
                        const Hop m = hop(g, cur, turn, tstep, origin, seen);
                        if (!m.ok || ++guard > 4 * h * w) {
                            err = "next move should be of len 1. Reached ambiguous situation.";
                            return -22;
                        }
                        cur = m.next;
                        if (turn == 2 && m.progressed) ++tstep;
                        if (m.completed) {
                            ++nturns;
                            turn = t2;
                            tstep = 0;
                            origin = cur;
                        }
                        if (nturns == 2) turn = 0;
                        path.push_back(cur);
                    }
                    paths.push_back(path);
                    if (nturns == 1) break;  // traffic_helper.py:205-207
                }
            }
            routes.push_back(paths);
        }
    }

    route_off.assign(1, 0);
    route_rc.clear();
    size_t rpa = routes.empty() ? 0 : routes[0].size();
    for (const auto& per_arrival : routes) {
        if (per_arrival.size() != rpa) { err = "arrival points have unequal route counts"; return -22; }
        for (const auto& p : per_arrival) {
            for (const Cell& c : p) {
                route_rc.push_back(c.first);
                route_rc.push_back(c.second);
            }
            route_off.push_back((int32_t)(route_rc.size() / 2));
        }
    }
    if ((int)route_off.size() - 1 != npath) { err = "len(paths) != npath"; return -22; }  // :520
    if (road_out) road_out->assign(g.road.begin(), g.road.end());   // 0/1 road flags (the 'scalar' vocab grid, TJ:301-307)
    *h_out = h;
    *w_out = w;
    *base_out = base;
    *npath_out = npath;
    *narrival_out = (int)routes.size();
    *rpa_out = (int)rpa;
    return 0;
}

}  // namespace ic3
