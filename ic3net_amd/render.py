"""What the reference's `render()` puts on the screen, as data: the list of `(row, x, text, color_pair)` draw calls of
predator_prey_env.py:307-336 / traffic_junction_env.py:254-292 for ONE env, computed from a state readback (no curses:
the batched envs live on the GPU and a terminal UI is out of scope — DESIGN.md §9).  `cells_to_text` lays the calls
out as the text block a terminal would show.  Pinned to the reference's own drawing by tests/golden/render_fixture.json
(tests/test_render_cpu.py)."""

# colour pairs as the reference initialises them (predator_prey_env.py:49-52, traffic_junction_env.py:50-54)
PP_PREDATOR, PP_PREY, PP_BOTH, PP_EMPTY = 1, 2, 3, 4
TJ_GAS, TJ_CRASH, TJ_ROAD, TJ_BRAKE = 1, 2, 4, 5


def pp_cells(loc_r, loc_c, npredator, dim):
    """Entities 0..npredator-1 are predators, the rest prey.  A cell's label is one 'X' per predator followed by one 'P'
    per prey on it (PP:311-321), centred in 3 columns at x = 4 * column; '0' where nobody stands."""
    label = [['' for _ in range(dim)] for _ in range(dim)]
    for i, (r, c) in enumerate(zip(loc_r, loc_c)):
        label[int(r)][int(c)] += 'X' if i < npredator else 'P'
    calls = []
    for r in range(dim):
        for c in range(dim):
            item = label[r][c]
            if item:
                pair = PP_BOTH if ('X' in item and 'P' in item) else PP_PREDATOR if 'X' in item else PP_PREY
                calls.append([r, 4 * c, item.center(3), pair])
            else:
                calls.append([r, 4 * c, '0'.center(3), PP_EMPTY])
    calls.append([dim, 0, '\n', 0])
    return calls


def tj_cells(grid, outside_class, loc_r, loc_c, last_act):
    """`grid`: the env's (h, w) id grid; every car slot is drawn at its location (dead cars are parked on (0, 0), which
    the reference never draws, TJ:275-276).  A car that last accelerated adds '<>', one that braked '<b>' (TJ:261-271);
    the 'b's are dropped on screen and only pick the colour: lone braking car 5, lone moving car 1, several cars 2."""
    h, w = len(grid), len(grid[0])
    label = [['_' if grid[r][c] != outside_class else '' for c in range(w)] for r in range(h)]
    for r, c, act in zip(loc_r, loc_c, last_act):
        r, c = int(r), int(c)
        label[r][c] = label[r][c].replace('_', '') + ('<>' if int(act) == 0 else '<b>')
    calls = []
    for r in range(h):
        for c in range(w):
            if r == 0 and c == 0:
                continue
            item = label[r][c]
            if item == '_':
                calls.append([r, 4 * c, '_'.center(3), TJ_ROAD])
            elif '<>' in item and len(item) > 3:
                calls.append([r, 4 * c, item.replace('b', '').center(3), TJ_CRASH])
            elif '<>' in item:
                calls.append([r, 4 * c, item.center(3), TJ_GAS])
            elif 'b' in item and len(item) > 3:
                calls.append([r, 4 * c, item.replace('b', '').center(3), TJ_CRASH])
            elif 'b' in item:
                calls.append([r, 4 * c, item.replace('b', '').center(3), TJ_BRAKE])
            else:
                calls.append([r, 4 * c, item.center(3), TJ_CRASH])
    calls.append([h, 0, '\n', 0])
    return calls


def cells_to_text(calls):
    """The draw calls on a blank character grid (later calls overwrite earlier ones, like a terminal)."""
    rows = {}
    for r, x, text, _ in calls:
        if text == '\n':
            continue
        line = rows.setdefault(r, [])
        if len(line) < x + len(text):
            line.extend(' ' * (x + len(text) - len(line)))
        line[x:x + len(text)] = list(text)
    return "\n".join("".join(rows.get(r, [])).rstrip() for r in range(max(rows) + 1 if rows else 0))
