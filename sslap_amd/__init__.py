"""sslap_amd -- MI355X-native auction solver for sparse linear assignment problems.

Drop-in for the hot path of OllieBoyne/sslap: `auction_solve`, `from_matrix` / `from_sparse`
(the reference's `_from_matrix` / `_from_sparse`), `AuctionSolver` and the feasibility guard `hopcroft_solve`.  Everything computes on the GPU
through libmisslap.so (hand-written HIP for gfx950, C ABI in include/misslap.h); importing the
package never touches the GPU, but every solver call raises if the library or the GPU is missing.
"""
from .auction_solve import AuctionSolver, auction_solve, from_matrix, from_sparse, _from_matrix, _from_sparse
from .check_feasible import hopcroft_solve

__version__ = "0.1.0"
solve_batch = AuctionSolver.solve_batch  # many problems with the same number of persons in lockstep on one GPU (include/misslap.h: misslap_solve_batch)
__all__ = ["auction_solve", "hopcroft_solve", "from_matrix", "from_sparse", "AuctionSolver", "solve_batch"]
