/* oracle/cref_undef_e.h -- TEST INFRASTRUCTURE: undefines the coordinate-field macros of cref_ec.h */
#undef EF
#undef ET
#undef E_add
#undef E_sub
#undef E_mul
#undef E_sqr
#undef E_dbl
#undef E_neg
#undef E_inv
#undef E_is_zero
#undef E_eq
#undef E_one
#undef E_zero
