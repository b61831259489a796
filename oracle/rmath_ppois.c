/*
 * oracle/rmath_ppois.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the one piece of third-party arithmetic on the dada() hot
 * path that is NOT under /root/reference: R's libRmath  ppois(x, lambda,
 * lower_tail=FALSE, log_p=FALSE), reached from the reference at
 *   src/pval.cpp:50   Rcpp::ppois(n_repeats, E_reads, false)
 * (call chain src/pval.cpp:44-64 calc_pA <- :67-89 get_pA, src/Rmain.cpp:247,
 *  src/error.cpp:118).
 *
 * Dependency: base R "nmath" (ppois.c, pgamma.c, dpois.c, bd0.c, stirlerr.c,
 * pnorm.c, dnorm.c).  DESCRIPTION:18 only says R (>= 3.4.0) — unpinned; the
 * reference's own reading copy is pinned in a source comment at r-source commit
 * af7f52f70101960861e5d995d3a4bec010bc89e6 (src/pval.cpp:201-202), and the
 * skeleton of ppois/pgamma/pgamma_raw/dpois_wrap is reproduced in the comment
 * block src/pval.cpp:230-339.  R's sources are not available offline, so this
 * file restates the PUBLISHED algorithm (Morten Welinder's pgamma, Catherine
 * Loader's dpois_raw/bd0/stirlerr "saddle-point" form used through R 4.0.x,
 * W. J. Cody's 1969 pnorm) in the operation order of that era's nmath.
 *
 * PARITY STATUS: "parity unpinned" against genuine libRmath bits — the
 * reference ships no test vectors for this call and R is not installed here.
 * It is pinned numerically instead: tests/test_oracle.py::test_ppois_against_truth_and_scipy
 * checks it against the committed 60-digit mpmath grid (tests/golden/ppois_grid.npy, made by
 * tests/golden/make_golden.py) and scipy.special.pdtrc over all four pgamma regimes and the
 * 1e-292..1e-323 underflow band (<= 1e-13 relative).  lgammafn() is taken from
 * the C library's lgamma() (R's own Chebyshev version differs by <= 1-2 ulp).
 */
#include <math.h>
#include <float.h>

#ifndef M_LN2
#define M_LN2 0.693147180559945309417232121458
#endif
#define R_M_2PI          6.283185307179586476925286766559
#define R_M_LN_SQRT_2PI  0.918938533204672741780329736406
#define R_M_1_SQRT_2PI   0.398942280401432677939946059934
#define R_M_SQRT_32      5.656854249492380195206754896838

static const double scalefactor = 1.157920892373162e+77; /* (2^32)^8 = 2^256 */
/* If |x| > |k| * M_cutoff, then log[ exp(-x) * k^x ] =~= -x  (pgamma.c) */
static const double M_cutoff = M_LN2 * DBL_MAX_EXP / DBL_EPSILON;

static double R_Log1_Exp(double x) { return x > -M_LN2 ? log(-expm1(x)) : log1p(-exp(x)); }

/* ---- stirlerr.c : log(n!) - log( sqrt(2*pi*n)*(n/e)^n ) ------------------- */
static double o_stirlerr(double n)
{
    static const double S0 = 0.083333333333333333333;       /* 1/12 */
    static const double S1 = 0.00277777777777777777778;     /* 1/360 */
    static const double S2 = 0.00079365079365079365079365;  /* 1/1260 */
    static const double S3 = 0.000595238095238095238095238; /* 1/1680 */
    static const double S4 = 0.0008417508417508417508417508;/* 1/1188 */
    /* exact values for n = 0, 0.5, 1.0, ..., 15.0 (regenerated with mpmath at
       60 digits from lgamma(n+1)-(n+.5)log(n)+n-log(sqrt(2pi))) */
    static const double sferr_halves[31] = {
        0.0,
        0.1534264097200273452913848,   0.0810614667953272582196702,
        0.0548141210519176538961390,   0.0413406959554092940938221,
        0.03316287351993628748511048,  0.02767792568499833914878929,
        0.02374616365629749597132920,  0.02079067210376509311152277,
        0.01848845053267318523077934,  0.01664469118982119216319487,
        0.01513497322191737887351255,  0.01387612882307074799874573,
        0.01281046524292022692424986,  0.01189670994589177009505572,
        0.01110455975820691732662991,  0.010411265261972096497478567,
        0.009799416126158803298389475, 0.009255462182712732917728637,
        0.008768700134139385462952823, 0.008330563433362871256469318,
        0.007934114564314020547248100, 0.007573675487951840794972024,
        0.007244554301320383179543912, 0.006942840107209529865664152,
        0.006665247032707682442354394, 0.006408994188004207068439631,
        0.006171712263039457647532867, 0.005951370112758847735624416,
        0.005746216513010115682023589, 0.005554733551962801371038690
    };
    double nn;
    if (n <= 15.0) {
        nn = n + n;
        if (nn == (int)nn) return sferr_halves[(int)nn];
        return lgamma(n + 1.) - (n + 0.5) * log(n) + n - R_M_LN_SQRT_2PI;
    }
    nn = n * n;
    if (n > 500) return (S0 - S1 / nn) / n;
    if (n > 80)  return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 35)  return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}

/* ---- bd0.c : x log(x/np) + np - x, accurately for x ~ np ------------------ */
static double o_bd0(double x, double np)
{
    double ej, s, s1, v;
    int j;
    if (fabs(x - np) < 0.1 * (x + np)) {
        v = (x - np) / (x + np);
        s = (x - np) * v;
        if (fabs(s) < DBL_MIN) return s;
        ej = 2 * x * v;
        v = v * v;
        for (j = 1; j < 1000; j++) {
            ej *= v;
            s1 = s + ej / ((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * log(x / np) + np - x;
}

/* ---- dpois.c : dpois_raw (R <= 4.0.x form) -------------------------------- */
static double o_dpois_raw(double x, double lambda, int give_log)
{
    if (lambda == 0) return (x == 0) ? (give_log ? 0. : 1.) : (give_log ? -INFINITY : 0.);
    if (!isfinite(lambda)) return give_log ? -INFINITY : 0.;
    if (x < 0) return give_log ? -INFINITY : 0.;
    if (x <= lambda * DBL_MIN) return give_log ? -lambda : exp(-lambda);
    if (lambda < x * DBL_MIN) {
        double v = -lambda + x * log(lambda) - lgamma(x + 1);
        return give_log ? v : exp(v);
    }
    {
        double f = R_M_2PI * x, e = -o_stirlerr(x) - o_bd0(x, lambda);
        return give_log ? -0.5 * log(f) + e : exp(e) / sqrt(f);
    }
}

/* ---- pgamma.c helpers ------------------------------------------------------ */
static double o_logcf(double x, double i, double d, double eps)
{
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
        if (fabs(b2) > scalefactor) {
            a1 /= scalefactor; b1 /= scalefactor; a2 /= scalefactor; b2 /= scalefactor;
        } else if (fabs(b2) < 1 / scalefactor) {
            a1 *= scalefactor; b1 *= scalefactor; a2 *= scalefactor; b2 *= scalefactor;
        }
    }
    return a2 / b2;
}

static double o_log1pmx(double x)
{
    static const double minLog1Value = -0.79149064;
    if (x > 1 || x < minLog1Value) return log1p(x) - x;
    {
        double r = x / (2 + x), y = r * r;
        if (fabs(x) < 1e-2) {
            static const double two = 2;
            return r * ((((two / 9 * y + two / 7) * y + two / 5) * y + two / 3) * y - x);
        }
        return r * (2 * y * o_logcf(y, 3, 2, 1e-14) - x);
    }
}

/* lgamma1p(a) = log(gamma(a+1)); on this path a = alph is an integer >= 1, so
   only the |a| >= 0.5 arm of R's lgamma1p is reachable. */
static double o_lgamma1p(double a) { return lgamma(a + 1); }

static double o_dpois_wrap(double x_plus_1, double lambda, int give_log)
{
    if (x_plus_1 > 1) return o_dpois_raw(x_plus_1 - 1, lambda, give_log);
    if (lambda > fabs(x_plus_1 - 1) * M_cutoff) {
        double v = -lambda - lgamma(x_plus_1);
        return give_log ? v : exp(v);
    }
    {
        double d = o_dpois_raw(x_plus_1, lambda, give_log);
        return give_log ? d + log(x_plus_1 / lambda) : d * (x_plus_1 / lambda);
    }
}

static double o_pgamma_smallx(double x, double alph, int lower_tail, int log_p)
{
    double sum = 0, c = alph, n = 0, term;
    do {
        n++;
        c *= -x / n;
        term = c / (alph + n);
        sum += term;
    } while (fabs(term) > DBL_EPSILON * fabs(sum));
    if (lower_tail) {
        double f1 = log_p ? log1p(sum) : 1 + sum, f2;
        if (alph > 1) {
            f2 = o_dpois_raw(alph, x, log_p);
            f2 = log_p ? f2 + x : f2 * exp(x);
        } else if (log_p)
            f2 = alph * log(x) - o_lgamma1p(alph);
        else
            f2 = pow(x, alph) / exp(o_lgamma1p(alph));
        return log_p ? f1 + f2 : f1 * f2;
    } else {
        double lf2 = alph * log(x) - o_lgamma1p(alph);
        if (log_p) return R_Log1_Exp(log1p(sum) + lf2);
        {
            double f1m1 = sum, f2m1 = expm1(lf2);
            return -(f1m1 + f2m1 + f1m1 * f2m1);
        }
    }
}

static double o_pd_upper_series(double x, double y, int log_p)
{
    double term = x / y, sum = term;
    do {
        y++;
        term *= x / y;
        sum += term;
    } while (term > sum * DBL_EPSILON);
    return log_p ? log(sum) : sum;
}

static double o_pd_lower_cf(double y, double d)
{
    double f = 0.0, of, f0, i, c2, c3, c4, a1, b1, a2, b2;
    if (y == 0) return 0;
    f0 = y / d;
    if (fabs(y - 1) < fabs(d) * DBL_EPSILON) return f0;
    if (f0 > 1.) f0 = 1.;
    c2 = y; c4 = d;
    a1 = 0; b1 = 1; a2 = y; b2 = d;
    while (b2 > scalefactor) { a1 /= scalefactor; b1 /= scalefactor; a2 /= scalefactor; b2 /= scalefactor; }
    i = 0; of = -1.;
    while (i < 200000) {
        i++; c2--; c3 = i * c2; c4 += 2;
        a1 = c4 * a2 + c3 * a1;
        b1 = c4 * b2 + c3 * b1;
        i++; c2--; c3 = i * c2; c4 += 2;
        a2 = c4 * a1 + c3 * a2;
        b2 = c4 * b1 + c3 * b2;
        if (b2 > scalefactor) { a1 /= scalefactor; b1 /= scalefactor; a2 /= scalefactor; b2 /= scalefactor; }
        if (b2 != 0) {
            f = a2 / b2;
            if (fabs(f - of) <= DBL_EPSILON * fmax(f0, fabs(f))) return f;
            of = f;
        }
    }
    return f;
}

static double o_pd_lower_series(double lambda, double y)
{
    double term = 1, sum = 0;
    while (y >= 1 && term > sum * DBL_EPSILON) {
        term *= y / lambda;
        sum += term;
        y--;
    }
    if (y != floor(y)) {
        double f = o_pd_lower_cf(y, lambda + 1 - y);
        sum += term * f;
    }
    return sum;
}

/* ---- pnorm.c (Cody 1969) / dnorm.c ---------------------------------------- */
static void o_pnorm_both(double x, double *cum, double *ccum, int i_tail, int log_p)
{
    static const double a[5] = { 2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
                                 18154.981253343561249, 0.065682337918207449113 };
    static const double b[4] = { 47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
                                 45507.789335026729956 };
    static const double c[9] = { 0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                                 597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326,
                                 11602.651437647350124, 9842.7148383839780218, 1.0765576773720192317e-8 };
    static const double d[8] = { 22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
                                 6485.558298266760755, 18615.571640885098091, 34900.952721145977266,
                                 38912.003286093271411, 19685.429676859990727 };
    static const double p[6] = { 0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
                                 0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303 };
    static const double q[5] = { 1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
                                 0.00378239633202758244, 7.29751555083966205e-5 };
    double xden, xnum, temp, del, eps, xsq, y;
    int i, lower, upper;
    eps = DBL_EPSILON * 0.5;
    lower = i_tail != 1;
    upper = i_tail != 0;
    y = fabs(x);
    if (y <= 0.67448975) {
        if (y > eps) {
            xsq = x * x;
            xnum = a[4] * xsq;
            xden = xsq;
            for (i = 0; i < 3; ++i) { xnum = (xnum + a[i]) * xsq; xden = (xden + b[i]) * xsq; }
        } else xnum = xden = 0.0;
        temp = x * (xnum + a[3]) / (xden + b[3]);
        if (lower) *cum = 0.5 + temp;
        if (upper) *ccum = 0.5 - temp;
        if (log_p) { if (lower) *cum = log(*cum); if (upper) *ccum = log(*ccum); }
    } else if (y <= R_M_SQRT_32) {
        xnum = c[8] * y;
        xden = y;
        for (i = 0; i < 7; ++i) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
        temp = (xnum + c[7]) / (xden + d[7]);
#define O_DO_DEL(X)                                                            \
        xsq = trunc((X) * 16) / 16;                                            \
        del = ((X) - xsq) * ((X) + xsq);                                       \
        if (log_p) {                                                           \
            *cum = (-xsq * xsq * 0.5) + (-del * 0.5) + log(temp);              \
            if ((lower && x > 0.) || (upper && x <= 0.))                       \
                *ccum = log1p(-exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp);\
        } else {                                                               \
            *cum = exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp;             \
            *ccum = 1.0 - *cum;                                                \
        }
#define O_SWAP_TAIL                                                            \
        if (x > 0.) { temp = *cum; if (lower) *cum = *ccum; *ccum = temp; }
        O_DO_DEL(y);
        O_SWAP_TAIL;
    } else if ((log_p && y < 1e170) || (lower && -37.5193 < x && x < 8.2924) ||
               (upper && -8.2924 < x && x < 37.5193)) {
        xsq = 1.0 / (x * x);
        xnum = p[5] * xsq;
        xden = xsq;
        for (i = 0; i < 4; ++i) { xnum = (xnum + p[i]) * xsq; xden = (xden + q[i]) * xsq; }
        temp = xsq * (xnum + p[4]) / (xden + q[4]);
        temp = (R_M_1_SQRT_2PI - temp) / y;
        O_DO_DEL(x);
        O_SWAP_TAIL;
    } else {
        if (x > 0) { *cum = log_p ? 0. : 1.; *ccum = log_p ? -INFINITY : 0.; }
        else       { *cum = log_p ? -INFINITY : 0.; *ccum = log_p ? 0. : 1.; }
    }
}

static double o_pnorm(double x, int lower_tail, int log_p)
{
    double p, cp;
    o_pnorm_both(x, &p, &cp, lower_tail ? 0 : 1, log_p);
    return lower_tail ? p : cp;
}

static double o_dnorm(double x, int give_log)
{
    x = fabs(x);
    if (x >= 2 * sqrt(DBL_MAX)) return give_log ? -INFINITY : 0.;
    if (give_log) return -(R_M_LN_SQRT_2PI + 0.5 * x * x);
    if (x < 5) return R_M_1_SQRT_2PI * exp(-0.5 * x * x);
    if (x > sqrt(-2 * M_LN2 * (DBL_MIN_EXP + 1 - DBL_MANT_DIG))) return 0.;
    {
        double x1 = ldexp(nearbyint(ldexp(x, 16)), -16), x2 = x - x1;
        return R_M_1_SQRT_2PI * (exp(-0.5 * x1 * x1) * exp((-0.5 * x2 - x1) * x2));
    }
}

static double o_dpnorm(double x, int lower_tail, double lp)
{
    if (x < 0) { x = -x; lower_tail = !lower_tail; }
    if (x > 10 && !lower_tail) {
        double term = 1 / x, sum = term, x2 = x * x, i = 1;
        do { term *= -i / x2; sum += term; i += 2; } while (fabs(term) > DBL_EPSILON * sum);
        return 1 / sum;
    }
    return o_dnorm(x, 0) / exp(lp);
}

static double o_ppois_asymp(double x, double lambda, int lower_tail, int log_p)
{
    static const double coefs_a[8] = { -1e99, 2 / 3., -4 / 135., 8 / 2835., 16 / 8505., -8992 / 12629925.,
                                       -334144 / 492567075., 698752 / 1477701225. };
    static const double coefs_b[8] = { -1e99, 1 / 12., 1 / 288., -139 / 51840., -571 / 2488320.,
                                       163879 / 209018880., 5246819 / 75246796800.,
                                       -534703531 / 902961561600. };
    double elfb, elfb_term, res12, res1_term, res1_ig, res2_term, res2_ig, dfm, pt_, s2pt, f, np;
    int i;
    dfm = lambda - x;
    pt_ = -o_log1pmx(dfm / x);
    s2pt = sqrt(2 * x * pt_);
    if (dfm < 0) s2pt = -s2pt;
    res12 = 0;
    res1_ig = res1_term = sqrt(x);
    res2_ig = res2_term = s2pt;
    for (i = 1; i < 8; i++) {
        res12 += res1_ig * coefs_a[i];
        res12 += res2_ig * coefs_b[i];
        res1_term *= pt_ / i;
        res2_term *= 2 * pt_ / (2 * i + 1);
        res1_ig = res1_ig / x + res1_term;
        res2_ig = res2_ig / x + res2_term;
    }
    elfb = x;
    elfb_term = 1;
    for (i = 1; i < 8; i++) { elfb += elfb_term * coefs_b[i]; elfb_term /= x; }
    if (!lower_tail) elfb = -elfb;
    f = res12 / elfb;
    np = o_pnorm(s2pt, !lower_tail, log_p);
    if (log_p) {
        double n_d_over_p = o_dpnorm(s2pt, !lower_tail, np);
        return np + log1p(f * n_d_over_p);
    }
    return np + f * o_dnorm(s2pt, 0);
}

/* pgamma_raw: structure exactly as reproduced at reference src/pval.cpp:259-319 */
static double o_pgamma_raw(double x, double alph, int lower_tail, int log_p)
{
    double res;
    if (x <= 0.) return lower_tail ? (log_p ? -INFINITY : 0.) : (log_p ? 0. : 1.);
    if (x >= INFINITY) return lower_tail ? (log_p ? 0. : 1.) : (log_p ? -INFINITY : 0.);
    if (x < 1) {
        res = o_pgamma_smallx(x, alph, lower_tail, log_p);
    } else if (x <= alph - 1 && x < 0.8 * (alph + 50)) {
        double sum = o_pd_upper_series(x, alph, log_p);
        double d = o_dpois_wrap(alph, x, log_p);
        if (!lower_tail) res = log_p ? R_Log1_Exp(d + sum) : 1 - d * sum;
        else res = log_p ? sum + d : sum * d;
    } else if (alph - 1 < x && alph < 0.8 * (x + 50)) {
        double sum, d = o_dpois_wrap(alph, x, log_p);
        if (alph < 1) {
            if (x * DBL_EPSILON > 1 - alph) sum = log_p ? 0. : 1.;
            else {
                double f = o_pd_lower_cf(alph, x - (alph - 1)) * x / alph;
                sum = log_p ? log(f) : f;
            }
        } else {
            sum = o_pd_lower_series(x, alph - 1);
            sum = log_p ? log1p(sum) : 1 + sum;
        }
        if (!lower_tail) res = log_p ? sum + d : sum * d;
        else res = log_p ? R_Log1_Exp(d + sum) : 1 - d * sum;
    } else {
        res = o_ppois_asymp(alph - 1, x, !lower_tail, log_p);
    }
    if (!log_p && res < DBL_MIN / DBL_EPSILON) return exp(o_pgamma_raw(x, alph, lower_tail, 1));
    return res;
}

/* ppois(x, lambda, lower_tail, log_p=FALSE)  — ppois.c; reference sketch at
   src/pval.cpp:240-247.  On the hot path lower_tail is always FALSE. */
double dada2_oracle_ppois(double x, double lambda, int lower_tail)
{
    if (isnan(x) || isnan(lambda)) return x + lambda;
    if (lambda < 0.) return NAN;
    if (x < 0) return lower_tail ? 0. : 1.;
    if (lambda == 0.) return lower_tail ? 1. : 0.;
    if (!isfinite(x)) return lower_tail ? 1. : 0.;
    x = floor(x + 1e-7);
    return o_pgamma_raw(lambda, x + 1, !lower_tail, 0);
}
