// bench/matmult/summa_gemm.cpp -- the reference's SUMMA bench (bench/matmult/summa_gemm.cpp:7-54) on MI355X.
// Same positional arguments and generators (A: distribute_random keyed rank/c, B and C keyed -(rank/c)), same protocol
// (num_iter invocations of matmult::summa::invoke between barriers); additionally prints time and TFLOP/s per
// iteration, which the reference leaves to its critter profiler.
//   summa_gemm <M> <N> <K> <rep_div> <layout> <num_chunks> <num_iter>
#include <cmath>
#include <iostream>

#include "../../src/alg/matmult/summa/summa.h"
#include "../launch.h"

int main(int argc, char** argv) {
  using T = double; using U = int64_t; using MatrixTypeR = matrix<T, U, rect>;
  if (argc < 8) {
    std::cerr << "usage: summa_gemm M N K rep_div layout num_chunks num_iter\n";
    return 2;
  }
  int rank = 0, size = 1;
  capital_bench::init(rank, size);
  const U M = atol(argv[1]), N = atol(argv[2]), K = atol(argv[3]);
  const U rep_div = atoi(argv[4]);
  const size_t layout = atoi(argv[5]);
  const size_t num_chunks = atoi(argv[6]);
  const size_t num_iter = atoi(argv[7]);
  // the reference needs a cubic grid (c = ceil(cbrt(P)) / rep_div, :27-28); 2 and 4 GPUs get the 1x1x2 and 2x2x1 grids
  size_t rep_factor = (size_t)std::nearbyint(std::ceil(std::cbrt((double)size))) / (size_t)std::max<U>(rep_div, 1);
  if (size == 2) rep_factor = 2;
  if (size == 4) rep_factor = 1;
  {
    topo::square SquareTopo(capital::world(), rep_factor, layout, num_chunks);
    MatrixTypeR matA(K, M, SquareTopo.d, SquareTopo.d);      // (columns, rows): A is M x K
    MatrixTypeR matB(N, K, SquareTopo.d, SquareTopo.d);
    MatrixTypeR matC(N, M, SquareTopo.d, SquareTopo.d);
    blas::ArgPack_gemm<T> blasArgs(blas::Order::AblasColumnMajor, blas::Transpose::AblasNoTrans, blas::Transpose::AblasNoTrans, 1., 0.);
    matA.distribute_random(SquareTopo.x, SquareTopo.y, SquareTopo.d, SquareTopo.d, rank / SquareTopo.c);
    matB.distribute_random(SquareTopo.x, SquareTopo.y, SquareTopo.d, SquareTopo.d, rank / SquareTopo.c * (-1));
    matC.distribute_random(SquareTopo.x, SquareTopo.y, SquareTopo.d, SquareTopo.d, rank / SquareTopo.c * (-1));
    matmult::summa::invoke(matA, matB, matC, SquareTopo, blasArgs);     // warm-up (workspace, kernel attributes)
    capital::sync();
    for (size_t i = 0; i < num_iter; ++i) {
      capital_bench::barrier();
      const double t0 = capital_bench::wtime();
      matmult::summa::invoke(matA, matB, matC, SquareTopo, blasArgs);
      capital::sync();
      const double total_time = capital_bench::max_over_ranks(capital_bench::wtime() - t0);
      if (rank == 0)
        std::cout << "total time - " << total_time << "   (" << 2.0 * (double)M * (double)N * (double)K / total_time / 1e12
                  << " TFLOP/s, " << size << " GPU)" << std::endl;
    }
    // CAPITAL_BENCH_DUMP=<path>: this rank's local block of C (column-major doubles) for an elementwise check against A B
    if (const char* dump = getenv("CAPITAL_BENCH_DUMP")) {
      auto host = matC.to_host();
      FILE* f = fopen((std::string(dump) + "." + std::to_string(rank)).c_str(), "wb");
      if (!f || fwrite(host.data(), sizeof(double), host.size(), f) != host.size()) { std::cerr << "cannot write " << dump << "\n"; return 1; }
      fclose(f);
    }
  }
  capital::finalize();
  return 0;
}
