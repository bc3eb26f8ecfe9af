// alg/alg.h -- umbrella include of the host-side layer (reference src/alg/alg.h:13-20).
#ifndef CAPITAL_ALG_H_
#define CAPITAL_ALG_H_

class NoAcceleration;  // reference tag (alg.h:9); unused

#include "./../util/shared.h"
#include "./../blas/engine.h"
#include "./../lapack/engine.h"
#include "./../matrix/matrix.h"
#include "./../matrix/serialize.h"
#include "./../util/topology.h"
#include "./../util/util.h"

#endif  // CAPITAL_ALG_H_
