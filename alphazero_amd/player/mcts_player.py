"""Drop-in for AlphaZero.player.mcts_player (Player.think / Player.ack)."""
from ..search import mcts as MCTS


class Player:
    def __init__(self, eval_fun, game_config, ext_config):
        self._game_config = game_config
        self.mcts = MCTS.MCTSearch(eval_fun.eval, self._game_config, max_playout=ext_config['max_playout'],
                                   leaves_per_round=ext_config.get('leaves_per_round', 1))

    def think(self, state, dirichlet=False):
        move, probs = self.mcts.calc_move_with_probs(state, dirichlet)
        return move, probs

    def ack(self, move):
        self.mcts.update_with_move(move)
