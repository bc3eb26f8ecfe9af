// bench/cholesky/cholinv.cpp -- the reference's Cholesky bench (bench/cholesky/cholinv.cpp:8-71) on MI355X.
// Same positional arguments, same generator, same protocol (one warm-up factor(), then num_iter timed calls between
// barriers, "total time - <s>" per iteration on rank 0); additionally prints the algorithmic TFLOP/s (n^3/3) and the
// validator's residual, which the reference keeps commented out (:61-66).
//   cholinv <num_rows> <rep_div> <complete_inv> <split> <bcMultiplier> <layout> <num_chunks> <num_iter>
#include <cmath>
#include <iostream>

#include "../../src/alg/cholesky/cholinv/cholinv.h"
#include "../../test/cholesky/validate.h"
#include "../launch.h"

int main(int argc, char** argv) {
  using T = double; using U = int64_t; using MatrixType = matrix<T, U, rect>;
  if (argc < 9) {
    std::cerr << "usage: cholinv num_rows rep_div complete_inv split bcMultiplier layout num_chunks num_iter\n";
    return 2;
  }
  int rank = 0, size = 1;
  capital_bench::init(rank, size);
  const char dir = 'U';
  const U num_rows = atol(argv[1]);          // rows of the global matrix
  const U rep_div = atoi(argv[2]);           // divides the depth of the cubic grid
  const bool complete_inv = atoi(argv[3]);   // complete the inverse at the top level?
  const U split = atoi(argv[4]);             // split shift
  const U bcMultiplier = atoi(argv[5]);      // base-case depth factor
  const size_t layout = atoi(argv[6]);
  const size_t num_chunks = atoi(argv[7]);   // > 0: chunked SUMMA pipeline on a second stream
  const size_t num_iter = atoi(argv[8]);

  using cholesky_type = cholesky::cholinv<cholesky::policy::cholinv::Serialize, cholesky::policy::cholinv::SaveIntermediates,
                                          cholesky::policy::cholinv::NoReplication>;
  // the reference needs a cubic grid (c = ceil(cbrt(P)) / rep_div, :33-34); 2 and 4 GPUs get the 1x1x2 and 2x2x1 grids
  size_t rep_factor = (size_t)std::nearbyint(std::ceil(std::cbrt((double)size))) / (size_t)std::max<U>(rep_div, 1);
  if (size == 2) rep_factor = 2;
  if (size == 4) rep_factor = 1;
  {
    topo::square SquareTopo(capital::world(), rep_factor, layout, num_chunks);
    MatrixType A(num_rows, num_rows, SquareTopo.d, SquareTopo.d);
    A.distribute_symmetric(SquareTopo.x, SquareTopo.y, SquareTopo.d, SquareTopo.d, rank / SquareTopo.c, true);
    cholesky_type::info<T, U> pack(complete_inv, split, bcMultiplier, dir);
    cholesky_type::factor(A, pack, SquareTopo);   // warm-up
    for (size_t i = 0; i < num_iter; ++i) {
      capital_bench::barrier();
      const double t0 = capital_bench::wtime();
      cholesky_type::factor(A, pack, SquareTopo);
      capital::sync();
      const double total_time = capital_bench::max_over_ranks(capital_bench::wtime() - t0);
      if (rank == 0)
        std::cout << "total time - " << total_time << "   (" << (double)num_rows * num_rows * num_rows / 3.0 / total_time / 1e12
                  << " TFLOP/s algorithmic, " << size << " GPU)" << std::endl;
    }
    const double res = capital_bench::max_over_ranks(cholesky::validate<cholesky_type>::residual(A, pack, SquareTopo));
    if (rank == 0) std::cout << "residual - " << res << std::endl;
  }
  capital::finalize();
  return 0;
}
