"""RNG helpers with the reference's names and draw order (AlphaZero/search/math_helper.py)."""
import random


def random_variate_dirichlet(alpha, length):
    # un-normalised gamma(alpha, 1) draws, one per root child
    return [random.gammavariate(alpha, 1) for _ in range(length)]


def weighted_random_choice(li):
    p, w = zip(*li)
    return random.choices(p, weights=w)[0]
