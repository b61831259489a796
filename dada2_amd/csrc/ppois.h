// ppois.h — abundance p-value arithmetic of the product: calc_pA (reference src/pval.cpp:44-64)
// and the Poisson upper tail it needs, P(X > n-1 | E) = pgamma(E, n, lower=TRUE), following the
// published algorithm of R's nmath (pgamma.c / dpois.c / bd0.c / stirlerr.c / pnorm.c as of the
// r-source commit cited at reference src/pval.cpp:201-202; skeleton reproduced at :230-339).
// One source for both sides: the HIP kernel k_calc_pA (device) and the host driver.  Specialised
// to the domain that reaches it on this path: shape alph = reads is an integer >= 1, lower tail.
// Compiled with -ffp-contract=off so host and device evaluate the same IEEE operation sequence;
// they still differ by the last ulp of exp/log/lgamma (glibc vs ocml) — see DESIGN.md §6.
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define D2_HD __host__ __device__ inline
#else
#define D2_HD inline
#endif

namespace d2 {
namespace pp {

constexpr double LN2 = 0.693147180559945309417232121458;
constexpr double M_2PI_ = 6.283185307179586476925286766559;
constexpr double LN_SQRT_2PI = 0.918938533204672741780329736406;
constexpr double ONE_SQRT_2PI = 0.398942280401432677939946059934;
constexpr double SQRT_32 = 5.656854249492380195206754896838;
constexpr double SCALEFACTOR = 1.157920892373162e+77;  // 2^256
constexpr double EPS = DBL_EPSILON;

D2_HD double log1_exp(double x) { return x > -LN2 ? log(-expm1(x)) : log1p(-exp(x)); }

D2_HD double stirlerr(double n) {
  const double S0 = 0.083333333333333333333, S1 = 0.00277777777777777777778, S2 = 0.00079365079365079365079365,
               S3 = 0.000595238095238095238095238, S4 = 0.0008417508417508417508417508;
  const double sferr_halves[31] = {
      0.0, 0.1534264097200273452913848, 0.0810614667953272582196702, 0.0548141210519176538961390,
      0.0413406959554092940938221, 0.03316287351993628748511048, 0.02767792568499833914878929,
      0.02374616365629749597132920, 0.02079067210376509311152277, 0.01848845053267318523077934,
      0.01664469118982119216319487, 0.01513497322191737887351255, 0.01387612882307074799874573,
      0.01281046524292022692424986, 0.01189670994589177009505572, 0.01110455975820691732662991,
      0.010411265261972096497478567, 0.009799416126158803298389475, 0.009255462182712732917728637,
      0.008768700134139385462952823, 0.008330563433362871256469318, 0.007934114564314020547248100,
      0.007573675487951840794972024, 0.007244554301320383179543912, 0.006942840107209529865664152,
      0.006665247032707682442354394, 0.006408994188004207068439631, 0.006171712263039457647532867,
      0.005951370112758847735624416, 0.005746216513010115682023589, 0.005554733551962801371038690};
  double nn;
  if (n <= 15.0) {
    nn = n + n;
    if (nn == (int)nn) return sferr_halves[(int)nn];
    return lgamma(n + 1.) - (n + 0.5) * log(n) + n - LN_SQRT_2PI;
  }
  nn = n * n;
  if (n > 500) return (S0 - S1 / nn) / n;
  if (n > 80) return (S0 - (S1 - S2 / nn) / nn) / n;
  if (n > 35) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
  return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}

D2_HD double bd0(double x, double np) {
  if (fabs(x - np) < 0.1 * (x + np)) {
    double v = (x - np) / (x + np), s = (x - np) * v;
    if (fabs(s) < DBL_MIN) return s;
    double ej = 2 * x * v;
    v = v * v;
    for (int j = 1; j < 1000; j++) {
      ej *= v;
      double s1 = s + ej / ((j << 1) + 1);
      if (s1 == s) return s1;
      s = s1;
    }
  }
  return x * log(x / np) + np - x;
}

D2_HD double dpois_raw(double x, double lambda, bool give_log) {
  if (lambda == 0) return (x == 0) ? (give_log ? 0. : 1.) : (give_log ? -INFINITY : 0.);
  if (!(fabs(lambda) <= DBL_MAX)) return give_log ? -INFINITY : 0.;   // !R_FINITE
  if (x < 0) return give_log ? -INFINITY : 0.;
  if (x <= lambda * DBL_MIN) return give_log ? -lambda : exp(-lambda);
  if (lambda < x * DBL_MIN) {
    double v = -lambda + x * log(lambda) - lgamma(x + 1);
    return give_log ? v : exp(v);
  }
  double f = M_2PI_ * x, e = -stirlerr(x) - bd0(x, lambda);
  return give_log ? -0.5 * log(f) + e : exp(e) / sqrt(f);
}

D2_HD double logcf(double x, double i, double d, double eps) {
  double c1 = 2 * d, c2 = i + d, c4 = c2 + d, a1 = c2;
  double b1 = i * (c2 - i * x), b2 = d * d * x, a2 = c4 * c2 - b2;
  b2 = c4 * b1 - i * b2;
  while (fabs(a2 * b1 - a1 * b2) > fabs(eps * b1 * b2)) {
    double c3 = c2 * c2 * x;
    c2 += d; c4 += d;
    a1 = c4 * a2 - c3 * a1;
    b1 = c4 * b2 - c3 * b1;
    c3 = c1 * c1 * x;
    c1 += d; c4 += d;
    a2 = c4 * a1 - c3 * a2;
    b2 = c4 * b1 - c3 * b2;
    if (fabs(b2) > SCALEFACTOR) {
      a1 /= SCALEFACTOR; b1 /= SCALEFACTOR; a2 /= SCALEFACTOR; b2 /= SCALEFACTOR;
    } else if (fabs(b2) < 1 / SCALEFACTOR) {
      a1 *= SCALEFACTOR; b1 *= SCALEFACTOR; a2 *= SCALEFACTOR; b2 *= SCALEFACTOR;
    }
  }
  return a2 / b2;
}

D2_HD double log1pmx(double x) {
  const double minLog1Value = -0.79149064;
  if (x > 1 || x < minLog1Value) return log1p(x) - x;
  double r = x / (2 + x), y = r * r;
  if (fabs(x) < 1e-2) {
    const double two = 2;
    return r * ((((two / 9 * y + two / 7) * y + two / 5) * y + two / 3) * y - x);
  }
  return r * (2 * y * logcf(y, 3, 2, 1e-14) - x);
}

// dpois_wrap(x+1, lambda) = dpois(x, lambda); alph >= 1 integer here, so only the first two arms
// of R's dpois_wrap are reachable (x_plus_1 > 1, or x_plus_1 == 1).
D2_HD double dpois_wrap(double x_plus_1, double lambda, bool give_log) {
  if (x_plus_1 > 1) return dpois_raw(x_plus_1 - 1, lambda, give_log);
  // x_plus_1 == 1: lambda > |0| * M_cutoff is always true for lambda > 0
  double v = -lambda - lgamma(x_plus_1);
  return give_log ? v : exp(v);
}

D2_HD double pgamma_smallx_lower(double x, double alph, bool log_p) {
  double sum = 0, c = alph, n = 0, term;
  do {
    n++;
    c *= -x / n;
    term = c / (alph + n);
    sum += term;
  } while (fabs(term) > EPS * fabs(sum));
  double f1 = log_p ? log1p(sum) : 1 + sum, f2;
  if (alph > 1) {
    f2 = dpois_raw(alph, x, log_p);
    f2 = log_p ? f2 + x : f2 * exp(x);
  } else if (log_p)
    f2 = alph * log(x) - lgamma(alph + 1);
  else
    f2 = pow(x, alph) / exp(lgamma(alph + 1));
  return log_p ? f1 + f2 : f1 * f2;
}

D2_HD double pd_upper_series(double x, double y, bool log_p) {
  double term = x / y, sum = term;
  do {
    y++;
    term *= x / y;
    sum += term;
  } while (term > sum * EPS);
  return log_p ? log(sum) : sum;
}

// y is an integer here (alph - 1), so R's continued-fraction tail (y != floor(y)) is unreachable.
D2_HD double pd_lower_series(double lambda, double y) {
  double term = 1, sum = 0;
  while (y >= 1 && term > sum * EPS) {
    term *= y / lambda;
    sum += term;
    y--;
  }
  return sum;
}

D2_HD void pnorm_both(double x, double *cum, double *ccum, int i_tail, bool log_p) {
  const double a[5] = {2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582, 18154.981253343561249,
                       0.065682337918207449113};
  const double b[4] = {47.20258190468824187, 976.09855173777669322, 10260.932208618978205, 45507.789335026729956};
  const double c[9] = {0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                       597.27027639480026226,  2494.5375852903726711, 6848.1904505362823326,
                       11602.651437647350124,  9842.7148383839780218, 1.0765576773720192317e-8};
  const double d[8] = {22.266688044328115691, 235.38790178262499861, 1519.377599407554805,  6485.558298266760755,
                       18615.571640885098091, 34900.952721145977266, 38912.003286093271411, 19685.429676859990727};
  const double p[6] = {0.21589853405795699,     0.1274011611602473639, 0.022235277870649807,
                       0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
  const double q[5] = {1.28426009614491121, 0.468238212480865118, 0.0659881378689285515, 0.00378239633202758244,
                       7.29751555083966205e-5};
  double xden, xnum, temp, del, xsq, y;
  const double eps = DBL_EPSILON * 0.5;
  const bool lower = i_tail != 1, upper = i_tail != 0;
  y = fabs(x);
  if (y <= 0.67448975) {
    if (y > eps) {
      xsq = x * x;
      xnum = a[4] * xsq;
      xden = xsq;
      for (int i = 0; i < 3; ++i) { xnum = (xnum + a[i]) * xsq; xden = (xden + b[i]) * xsq; }
    } else xnum = xden = 0.0;
    temp = x * (xnum + a[3]) / (xden + b[3]);
    if (lower) *cum = 0.5 + temp;
    if (upper) *ccum = 0.5 - temp;
    if (log_p) { if (lower) *cum = log(*cum); if (upper) *ccum = log(*ccum); }
    return;
  }
  bool mid = y <= SQRT_32;
  bool far = !mid && ((log_p && y < 1e170) || (lower && -37.5193 < x && x < 8.2924) || (upper && -8.2924 < x && x < 37.5193));
  if (!mid && !far) {
    if (x > 0) { *cum = log_p ? 0. : 1.; *ccum = log_p ? -INFINITY : 0.; }
    else { *cum = log_p ? -INFINITY : 0.; *ccum = log_p ? 0. : 1.; }
    return;
  }
  double X;
  if (mid) {
    xnum = c[8] * y;
    xden = y;
    for (int i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
    temp = (xnum + c[7]) / (xden + d[7]);
    X = y;
  } else {
    xsq = 1.0 / (x * x);
    xnum = p[5] * xsq;
    xden = xsq;
    for (int i = 0; i < 4; ++i) { xnum = (xnum + p[i]) * xsq; xden = (xden + q[i]) * xsq; }
    temp = xsq * (xnum + p[4]) / (xden + q[4]);
    temp = (ONE_SQRT_2PI - temp) / y;
    X = x;
  }
  xsq = trunc(X * 16) / 16;
  del = (X - xsq) * (X + xsq);
  if (log_p) {
    *cum = (-xsq * xsq * 0.5) + (-del * 0.5) + log(temp);
    if ((lower && x > 0.) || (upper && x <= 0.)) *ccum = log1p(-exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp);
  } else {
    *cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;
    *ccum = 1.0 - *cum;
  }
  if (x > 0.) { temp = *cum; if (lower) *cum = *ccum; *ccum = temp; }
}

D2_HD double pnorm01(double x, bool lower_tail, bool log_p) {
  double p = 0, cp = 0;
  pnorm_both(x, &p, &cp, lower_tail ? 0 : 1, log_p);
  return lower_tail ? p : cp;
}

D2_HD double dnorm01(double x) {
  x = fabs(x);
  if (x >= 2 * sqrt(DBL_MAX)) return 0.;
  if (x < 5) return ONE_SQRT_2PI * exp(-0.5 * x * x);
  if (x > sqrt(-2 * LN2 * (DBL_MIN_EXP + 1 - DBL_MANT_DIG))) return 0.;
  double x1 = ldexp(nearbyint(ldexp(x, 16)), -16), x2 = x - x1;
  return ONE_SQRT_2PI * (exp(-0.5 * x1 * x1) * exp((-0.5 * x2 - x1) * x2));
}

D2_HD double dpnorm(double x, bool lower_tail, double lp) {
  if (x < 0) { x = -x; lower_tail = !lower_tail; }
  if (x > 10 && !lower_tail) {
    double term = 1 / x, sum = term, x2 = x * x, i = 1;
    do { term *= -i / x2; sum += term; i += 2; } while (fabs(term) > EPS * sum);
    return 1 / sum;
  }
  return dnorm01(x) / exp(lp);
}

D2_HD double ppois_asymp(double x, double lambda, bool lower_tail, bool log_p) {
  const double coefs_a[8] = {-1e99, 2 / 3., -4 / 135., 8 / 2835., 16 / 8505., -8992 / 12629925.,
                             -334144 / 492567075., 698752 / 1477701225.};
  const double coefs_b[8] = {-1e99, 1 / 12., 1 / 288., -139 / 51840., -571 / 2488320., 163879 / 209018880.,
                             5246819 / 75246796800., -534703531 / 902961561600.};
  double dfm = lambda - x;
  double pt_ = -log1pmx(dfm / x);
  double s2pt = sqrt(2 * x * pt_);
  if (dfm < 0) s2pt = -s2pt;
  double res12 = 0, res1_term, res1_ig, res2_term, res2_ig;
  res1_ig = res1_term = sqrt(x);
  res2_ig = res2_term = s2pt;
  for (int i = 1; i < 8; i++) {
    res12 += res1_ig * coefs_a[i];
    res12 += res2_ig * coefs_b[i];
    res1_term *= pt_ / i;
    res2_term *= 2 * pt_ / (2 * i + 1);
    res1_ig = res1_ig / x + res1_term;
    res2_ig = res2_ig / x + res2_term;
  }
  double elfb = x, elfb_term = 1;
  for (int i = 1; i < 8; i++) { elfb += elfb_term * coefs_b[i]; elfb_term /= x; }
  if (!lower_tail) elfb = -elfb;
  double f = res12 / elfb;
  double np = pnorm01(s2pt, !lower_tail, log_p);
  if (log_p) {
    double n_d_over_p = dpnorm(s2pt, !lower_tail, np);
    return np + log1p(f * n_d_over_p);
  }
  return np + f * dnorm01(s2pt);
}

// pgamma_raw(x, alph, lower_tail = TRUE, log_p)   (reference sketch src/pval.cpp:259-319)
D2_HD double pgamma_lower(double x, double alph, bool log_p) {
  double res;
  if (x <= 0.) return log_p ? -INFINITY : 0.;
  if (x >= INFINITY) return log_p ? 0. : 1.;
  if (x < 1) {
    res = pgamma_smallx_lower(x, alph, log_p);
  } else if (x <= alph - 1 && x < 0.8 * (alph + 50)) {
    double sum = pd_upper_series(x, alph, log_p);
    double d = dpois_wrap(alph, x, log_p);
    res = log_p ? sum + d : sum * d;
  } else if (alph - 1 < x && alph < 0.8 * (x + 50)) {
    double d = dpois_wrap(alph, x, log_p);
    double sum = pd_lower_series(x, alph - 1);
    sum = log_p ? log1p(sum) : 1 + sum;
    res = log_p ? log1_exp(d + sum) : 1 - d * sum;
  } else {
    res = ppois_asymp(alph - 1, x, false, log_p);
  }
  return res;
}

// ppois(x, lambda, lower_tail = FALSE, log_p = FALSE) for integer x >= 0
D2_HD double ppois_upper(double x, double lambda) {
  if (lambda < 0.) return NAN;
  if (x < 0) return 1.;
  if (lambda == 0.) return 0.;
  x = floor(x + 1e-7);
  double res = pgamma_lower(lambda, x + 1, false);
  // results this close to DBL_MIN are redone in log space (pgamma.c; reference src/pval.cpp:309-317)
  if (res < DBL_MIN / DBL_EPSILON) return exp(pgamma_lower(lambda, x + 1, true));
  return res;
}

// calc_pA — reference src/pval.cpp:44-64
D2_HD double calc_pA(int reads, double E_reads, bool prior) {
  double pval = ppois_upper((double)(reads - 1), E_reads);
  if (!prior) {
    double norm = 1.0 - exp(-E_reads);
    if (norm < 1e-7) norm = E_reads - 0.5 * E_reads * E_reads;  // TAIL_APPROX_CUTOFF, dada.h:25
    pval = pval / norm;
  }
  return pval;
}

}  // namespace pp
}  // namespace d2
