/* placeholder: SPA oracle lives here (filled in below) */
