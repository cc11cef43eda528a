"""Host logic of gating matches (train/parallel/evaluator.py:86-89,114-115,122): colour alternation and
the 55 % acceptance rule.  No GPU needed."""
from alphazero_amd.game.gameplay import gating_colors, gating_verdict

BLACK, WHITE = 1, -1


def reference_rule(winners, colors):
    # restated from evaluator.py:86-89 and :122
    win_counter = num_not_tie = 0
    for winner, color_of_new in zip(winners, colors):
        if winner == color_of_new:
            win_counter += 1
        if winner is not None and winner != 0:
            num_not_tie += 1
    return win_counter, num_not_tie, win_counter > int(0.55 * num_not_tie)


def test_colors_alternate_black_first():
    assert gating_colors(0) == []
    assert gating_colors(1) == [BLACK]
    assert gating_colors(4) == [BLACK, WHITE, BLACK, WHITE]
    assert gating_colors(5) == [BLACK, WHITE, BLACK, WHITE, BLACK]


def test_verdict_matches_reference_rule():
    import itertools
    for n in range(1, 7):
        colors = gating_colors(n)
        for winners in itertools.product([BLACK, WHITE, 0], repeat=n):
            assert gating_verdict(list(winners), colors) == reference_rule(winners, colors)
    # 10 games, 6 decisive wins out of 10: int(5.5) = 5 -> accepted; 5 of 10 -> not
    colors = gating_colors(10)
    assert gating_verdict(colors[:6] + [-c for c in colors[6:]], colors) == (6, 10, True)
    assert gating_verdict(colors[:5] + [-c for c in colors[5:]], colors) == (5, 10, False)
    # draws leave the denominator: 2 wins, 1 loss, 7 draws -> int(1.65) = 1 -> accepted
    assert gating_verdict(colors[:2] + [-colors[2]] + [0] * 7, colors) == (2, 3, True)
