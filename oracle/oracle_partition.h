// TEST INFRASTRUCTURE ONLY (see oracle_common.h).
#pragma once
#include <cstdint>
#include <vector>
#include "oracle_common.h"

namespace oracle {
std::vector<double> NegativeBinomialWrapper(double mean, double variance, int maxValue);
std::vector<std::vector<int>> GetGenotypeCombinations(int numberOfStates, int currentState);
int hmm_chromosome(int nSamples, bool perSample, const double* const* cov, int T, const double* medians,
                   const double* pseudoVariances, int32_t* path);
void hmm_global_params(int nchr, const double* const* cov, const int64_t* n, double* median, double* pseudoVariance);

struct CbsStats { int64_t tmaxo_calls = 0, tmaxo_elems = 0, perms = 0, perm_elems = 0, tpermp_draws = 0, tailp_exits = 0, big_t_splits = 0; };

void ComputeBoundary(uint32_t nPerm, double alpha, double eta, std::vector<uint32_t>& sbdry);
double TailP(double b, double delta, int m, int nGrid, double tol);
void TMaxO(const double* x, int n, double tss, double* sx, int iseg[2], double& ostat, int al0);
double HTMaxP(int k, double tss, const double* px, int n, double* sx, int al0);
double TMaxP(double tss, const double* px, int n, double* sx, int al0);
double phyper_lower(double x, double NR, double NB, double n);
// ChangePoints on one chromosome; returns segment lengths.
std::vector<int> ChangePoints(const double* genomeData, int n, const std::vector<uint32_t>& sbdry, MT19937& rnd, double alpha,
                              uint32_t nPerm, int minWidth, int kMax, uint32_t nMin, int undoSplits, double trimmedSD,
                              double undoPrune, double undoSD, CbsStats* stats);
// ChangePoint.ChangePointsPrune (ChangePoint.cs:205-271)
std::vector<int> ChangePointsPrune(const double* gd, int n, const std::vector<int>& lengthSeg, double changeCutoff);
double TrimmedVariance(const std::vector<const double*>& scores, const std::vector<int>& lens, double trim);
void dotnet_sort_keys_items(double* keys, int* items, int index, int length, int arrayLength);
}  // namespace oracle
