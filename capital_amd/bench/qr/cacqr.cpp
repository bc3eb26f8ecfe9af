// bench/qr/cacqr.cpp -- the reference's CholeskyQR bench (bench/qr/cacqr.cpp:8-77) on MI355X.  Same twelve positional
// arguments and loops (c from rep_factor_start..end, bc from bcMultiplier_start..end, num_iter warm-ups, one timed
// factor, "m n c bc seconds" on rank 0); additionally prints TFLOP/s (4mn^2 for variant 2) and the two validators.
//   cacqr <variant> <num_rows> <num_columns> <rep_start> <rep_end> <complete_inv> <split> <bc_start> <bc_end> <layout> <num_chunks> <num_iter>
#include <iostream>

#include "../../src/alg/qr/cacqr/cacqr.h"
#include "../../test/qr/validate.h"
#include "../launch.h"

int main(int argc, char** argv) {
  using T = double; using U = int64_t; using MatrixType = matrix<T, U, rect>;
  if (argc < 13) {
    std::cerr << "usage: cacqr variant num_rows num_columns rep_start rep_end complete_inv split bc_start bc_end layout num_chunks num_iter\n";
    return 2;
  }
  int rank = 0, size = 1;
  capital_bench::init(rank, size);
  const size_t variant = atoi(argv[1]);      // 1 CholeskyQR, 2 CholeskyQR2
  const U num_rows = atol(argv[2]);
  const U num_columns = atol(argv[3]);
  const U rep_factor_start = atoi(argv[4]), rep_factor_end = atoi(argv[5]);
  const bool complete_inv = atoi(argv[6]);
  const U split = atoi(argv[7]);
  const U bc_start = atoi(argv[8]), bc_end = atoi(argv[9]);
  const size_t layout = atoi(argv[10]), num_chunks = atoi(argv[11]), num_iter = atoi(argv[12]);

  using qr_type = qr::cacqr<qr::policy::cacqr::Serialize, qr::policy::cacqr::SaveIntermediates>;
  using ci_type = cholesky::cholinv<cholesky::policy::cholinv::Serialize, cholesky::policy::cholinv::SaveIntermediates,
                                    cholesky::policy::cholinv::NoReplication>;
  for (U i = rep_factor_start; i <= rep_factor_end; ++i) {
    topo::rect RectTopo(capital::world(), (size_t)i, layout, num_chunks);
    MatrixType A(num_columns, num_rows, RectTopo.c, RectTopo.d);
    A.distribute_random(RectTopo.x, RectTopo.y, RectTopo.c, RectTopo.d, rank / RectTopo.c);
    for (U j = bc_start; j <= bc_end; ++j) {
      ci_type::info<T, U> ci_pack(complete_inv, split, j, 'U');
      qr_type::info<T, U, ci_type> pack(variant, ci_pack);
      for (size_t k = 0; k < num_iter; ++k) {      // warm-ups regenerate the input, as the reference does (:43-46)
        A.distribute_random(RectTopo.x, RectTopo.y, RectTopo.c, RectTopo.d, rank / RectTopo.c);
        capital_bench::barrier();
        qr_type::factor(A, pack, RectTopo);
      }
      A.distribute_random(RectTopo.x, RectTopo.y, RectTopo.c, RectTopo.d, rank / RectTopo.c);
      capital_bench::barrier();
      const double t0 = capital_bench::wtime();
      qr_type::factor(A, pack, RectTopo);
      capital::sync();
      const double secs = capital_bench::max_over_ranks(capital_bench::wtime() - t0);
      const double res = capital_bench::max_over_ranks(qr::validate<qr_type>::residual(A, pack, RectTopo));
      const double orth = qr::validate<qr_type>::orthogonality(A, pack, RectTopo);
      if (rank == 0) {
        std::cout << num_rows << " " << num_columns << " " << i << " " << j << " " << secs << std::endl;
        std::cout << "  " << (variant == 2 ? 4.0 : 2.0) * num_rows * num_columns * num_columns / secs / 1e12 << " TFLOP/s algorithmic, residual "
                  << res << ", orthogonality " << orth << std::endl;
      }
    }
  }
  capital::finalize();
  return 0;
}
