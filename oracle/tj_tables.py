"""TEST INFRASTRUCTURE ONLY (oracle).  numpy restatement of the Traffic-Junction init-time tables.

Follows /root/reference/ic3net-envs/ic3net_envs/traffic_junction_env.py (TJ) `multi_agent_init`
:80-158, `_set_grid` :300-319, `_set_paths_easy` :395-410, `_set_paths` :509-523 and
traffic_helper.py (TH) `get_road_blocks` :5-22, `get_add_mat` :28-96, `next_move` :99-152,
`get_routes` :156-209.  Pinned by tests/golden/tj_tables_*.npz captured from the reference.

`build()` returns plain ints / int arrays:
  h, w, vocab, outside, car_class, base, npath, narrival, routes_per_arrival,
  grid (h,w) road ids, pad_grid, route_off (npath+1), route_rc (total,2), routes (list of arrays)
"""
import math

import numpy as np

NROAD = {'easy': 2, 'medium': 4, 'hard': 8}                      # TJ:117-119
STEPS = ((-1, 0), (1, 0), (0, -1), (0, 1))                        # TH:3 neighbour order


def road_slices(w, h, difficulty):
    """TH:5-22.  NB the last 'hard' slice uses h for a column range (sic, TH:20)."""
    if difficulty == 'easy':
        return [np.s_[h // 2, :], np.s_[:, w // 2]]
    if difficulty == 'medium':
        return [np.s_[h // 2 - 1:h // 2 + 1, :], np.s_[:, w // 2 - 1:w // 2 + 1]]
    return [np.s_[h // 3 - 2:h // 3, :], np.s_[2 * h // 3:2 * h // 3 + 2, :],
            np.s_[:, w // 3 - 2:w // 3], np.s_[:, 2 * h // 3:2 * h // 3 + 2]]


def id_grid(h, w, difficulty, outside):
    """TJ:300-314: road cells numbered slice by slice; later slices overwrite junction cells.
    Also returns the 0/1 route grid (TJ:309) the route walker uses.
    NB `w, h = self.dims` at TJ:302 (swapped names; dims are square)."""
    g = np.full((h, w), outside, dtype=np.int64)
    route = np.full((h, w), 0, dtype=np.int64)      # OUTSIDE_CLASS is 0 when route_grid is copied
    start = 0
    sl = road_slices(h, w, difficulty)
    for s in sl:
        route[s] = 1                                # ROAD_CLASS TJ:307
    for s in sl:
        n = int(np.prod(g[s].shape))
        g[s] = np.arange(start, start + n).reshape(g[s].shape)
        start += n
    return g, route


def _aux(h, w, route, difficulty):
    """TH:28-96 arrival/finish points, lane-direction map, junction map."""
    lane = route.copy()
    junc = np.zeros_like(route)
    if difficulty == 'medium':
        arrive = [(0, w // 2 - 1), (h - 1, w // 2), (h // 2, 0), (h // 2 - 1, w - 1)]
        finish = [(0, w // 2), (h - 1, w // 2 - 1), (h // 2 - 1, 0), (h // 2, w - 1)]
        lane[h // 2, :] = 2
        lane[h // 2 - 1, :] = 3
        lane[:, w // 2] = 4
        junc[h // 2 - 1:h // 2 + 1, w // 2 - 1:w // 2 + 1] = 1
    else:
        arrive = [(0, w // 3 - 2), (0, 2 * w // 3), (h // 3 - 1, 0), (2 * h // 3 + 1, 0),
                  (h - 1, w // 3 - 1), (h - 1, 2 * w // 3 + 1), (h // 3 - 2, w - 1), (2 * h // 3, w - 1)]
        finish = [(0, w // 3 - 1), (0, 2 * w // 3 + 1), (h // 3 - 2, 0), (2 * h // 3, 0),
                  (h - 1, w // 3 - 2), (h - 1, 2 * w // 3), (h // 3 - 1, w - 1), (2 * h // 3 + 1, w - 1)]
        lane[h // 3 - 1, :] = 2
        lane[2 * h // 3, :] = 3
        lane[2 * h // 3 + 1, :] = 4
        lane[:, w // 3 - 2] = 5
        lane[:, w // 3 - 1] = 6
        lane[:, 2 * w // 3] = 7
        lane[:, 2 * w // 3 + 1] = 8
        for r0 in (h // 3 - 2, 2 * h // 3):
            for c0 in (w // 3 - 2, 2 * w // 3):
                junc[r0:r0 + 2, c0:c0 + 2] = 1
    return arrive, finish, lane, junc


def _advance(cur, turn, turn_step, origin, route, lane, junc, seen):
    """TH:99-152 next_move: first admissible neighbour in STEPS order."""
    h, w = route.shape
    progressed = completed = False
    picks = []
    for dr, dc in STEPS:
        n = (cur[0] + dr, cur[1] + dc)
        if not (0 <= n[0] <= h - 1 and 0 <= n[1] <= w - 1) or not route[n] or n in seen:
            continue
        if junc[n] == junc[cur] == 1:                                           # TH:109-123
            if turn in (0, 2) and (n[0] == origin[0] or n[1] == origin[1]):
                picks.append(n)
                if turn == 2:
                    progressed = True
            elif turn == 2 and turn_step == 1:
                picks.append(n)
                progressed = True
        elif junc[cur] and not junc[n] and turn == 2 and turn_step == 2 \
                and (abs(origin[0] - n[0]) == 2 or abs(origin[1] - n[1]) == 2):     # TH:126-129
            picks.append(n)
            completed = True
        elif junc[n] and not junc[cur]:                                         # TH:132-133
            picks.append(n)
        elif turn == 1 and not junc[n] and junc[cur]:                           # TH:136-138
            picks.append(n)
            completed = True
        elif turn == 0 and junc[cur] and lane[n] == lane[origin]:               # TH:141-143
            picks.append(n)
            completed = True
        elif lane[n] == lane[cur] and not junc[cur]:                            # TH:146-147
            picks.append(n)
    if not picks:
        raise RuntimeError("next move should be of len 1. Reached ambiguous situation.")
    return picks[0], progressed, completed


def walk_routes(h, w, route, difficulty):
    """TH:156-209 get_routes."""
    arrive, finish, lane, junc = _aux(h, w, route, difficulty)
    second = 1 if difficulty == 'medium' else 3
    out = []
    for i, a in enumerate(arrive):
        goals = finish[:i] + finish[i + 1:]                                      # TH:24-25
        paths = []
        for t1 in range(3):
            for t2 in range(second):
                nturns, turn, tstep = 0, t1, 0
                cur = origin = a
                path, seen = [cur], set()
                while cur not in goals:
                    seen.add(cur)
                    cur, prog, done = _advance(cur, turn, tstep, origin, route, lane, junc, seen)
                    if turn == 2 and prog:
                        tstep += 1
                    if done:
                        nturns += 1
                        turn, tstep, origin = t2, 0, cur
                    if nturns == 2:
                        turn = 0
                    path.append(cur)
                paths.append(np.array(path, dtype=np.int64))
                if nturns == 1:                                                  # TH:205-207
                    break
        out.append(paths)
    return out


def build(dim, vision, difficulty, vocab_type='bool'):
    h = w = dim
    if difficulty in ('medium', 'easy'):                                         # TJ:93-96
        assert dim % 2 == 0, 'Only even dimension supported for now.'
        assert dim >= 4 + vision, 'Min dim: 4 + vision'
    if difficulty == 'hard':                                                     # TJ:98-100
        assert dim >= 9, 'Min dim: 9'
        assert dim % 3 == 0, 'Hard version works for multiple of 3. dim. only.'
    if difficulty == 'easy':                                                     # TJ:112-115
        h = w = dim + 1
    nroad = NROAD[difficulty]
    base = {'easy': 2 * dim, 'medium': 4 * dim, 'hard': 8 * dim}[difficulty]     # TJ:121-124 (orig dim)
    npath = math.factorial(nroad) // math.factorial(nroad - 2)                   # TJ:126
    outside, car_class, vocab = base, base + 2, base + 3                         # TJ:131-134
    grid, route = id_grid(h, w, difficulty, outside)
    pad = np.pad(grid, vision, 'constant', constant_values=outside)              # TJ:317
    if difficulty == 'easy':                                                     # TJ:395-410
        routes = [[np.array([(i, w // 2) for i in range(h)], dtype=np.int64)],
                  [np.array([(h // 2, i) for i in range(w)], dtype=np.int64)]]
    else:
        routes = walk_routes(h, w, route, difficulty)
    if vocab_type == 'scalar':                                                   # TJ:139-148, 300-307
        vocab, outside, car_class, base = 2, 0, 2, 0                             # classes are not shifted
        grid = route.copy()                                                      # 0 outside / 1 road (ROAD_CLASS)
        pad = np.pad(grid, vision, 'constant', constant_values=0)
    flat = [p for r in routes for p in r]
    assert len(flat) == npath                                                    # TJ:520
    off = np.zeros(npath + 1, np.int32)
    off[1:] = np.cumsum([len(p) for p in flat])
    return dict(h=h, w=w, vocab=vocab, outside=outside, car_class=car_class, base=base, npath=npath,
                narrival=len(routes), routes_per_arrival=len(routes[0]), grid=grid.astype(np.int32),
                pad_grid=pad.astype(np.int32), route_off=off,
                route_rc=np.concatenate(flat).astype(np.int32), routes=flat)
