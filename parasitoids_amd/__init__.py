"""parasitoids_amd -- MI355X-native drift-diffusion forward solver.

Drop-in for the hot path of mountaindust/Parasitoids behind its own seam:
`hip_lib.HipSolve` (<- cuda_lib.CudaSolve), `CalcSol.get_solutions` /
`get_populations`, `ParasitoidModel.prob_mass`, `Run.Params`.  Importing the
package does not touch the GPU; `hip_lib` raises ImportError without it.
"""
__version__ = '0.1.0'
